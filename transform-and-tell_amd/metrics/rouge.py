"""ROUGE-L as the evaluation tail uses it (scripts/compute_metrics.py:26,146-147 -> pycocoevalcap.rouge.Rouge, a
third-party package absent here: restated from its published algorithm - Lin 2004 ROUGE-L F-measure with
beta = 1.2 over whitespace tokens, best reference per side; parity unpinned, known answers in the tests)."""


def lcs_length(a, b):
    """Length of the longest common subsequence of two token lists."""
    if len(a) < len(b):
        a, b = b, a
    prev = [0] * (len(b) + 1)
    for x in a:
        cur = [0]
        for j, y in enumerate(b, 1):
            cur.append(prev[j - 1] + 1 if x == y else max(prev[j], cur[j - 1]))
        prev = cur
    return prev[-1]


class Rouge:
    def __init__(self, beta=1.2):
        self.beta = beta

    def calc_score(self, candidate, refs):
        """candidate: [sentence], refs: [sentence, ...] -> ROUGE-L F of the one candidate against its references."""
        assert len(candidate) == 1 and len(refs) > 0
        cand = candidate[0].split(' ')
        prec, rec = [], []
        for ref in refs:
            toks = ref.split(' ')
            lcs = lcs_length(toks, cand)
            prec.append(lcs / float(len(cand)))
            rec.append(lcs / float(len(toks)))
        p, r = max(prec), max(rec)
        if p == 0 or r == 0:
            return 0.0
        return ((1 + self.beta ** 2) * p * r) / float(r + self.beta ** 2 * p)
