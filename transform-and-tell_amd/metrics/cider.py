"""CIDEr as the evaluation tail uses it (scripts/compute_metrics.py:25,77,148,179 -> pycocoevalcap.cider.CiderScorer(n=4,
sigma=6.0), a third-party package absent here: restated from its published algorithm - Vedantam et al. 2015 with the
clipping and Gaussian length penalty that scorer applies; parity unpinned, known answers in the tests).

tf-idf vectors of the 1..n-grams (document frequency over the REFERENCE sets of the evaluated corpus, idf = ln(#sets)
- ln(max(1, df))), per n the clipped cosine similarity min(hyp, ref) . ref / (|hyp| |ref|), times
exp(-(len_hyp - len_ref)^2 / (2 sigma^2)) - the scorer measures "length" as the number of bigrams -, mean over n and
references, times 10."""
import math
from collections import defaultdict


def precook(sentence, n=4):
    words = sentence.split()
    counts = defaultdict(int)
    for k in range(1, n + 1):
        for i in range(len(words) - k + 1):
            counts[tuple(words[i:i + k])] += 1
    return counts


class CiderScorer:
    def __init__(self, n=4, sigma=6.0):
        self.n, self.sigma = n, sigma
        self.crefs, self.ctest = [], []

    def __iadd__(self, other):
        test, refs = other
        self.crefs.append([precook(r, self.n) for r in refs])
        self.ctest.append(precook(test, self.n))
        return self

    def _vec(self, counts, df, ref_len):
        vec = [defaultdict(float) for _ in range(self.n)]
        norm = [0.0] * self.n
        length = 0
        for ngram, tf in counts.items():
            idf = ref_len - math.log(max(1.0, df.get(ngram, 0.0)))
            k = len(ngram) - 1
            vec[k][ngram] = float(tf) * idf
            norm[k] += vec[k][ngram] ** 2
            if k == 1:
                length += tf
        return vec, [math.sqrt(x) for x in norm], length

    def _sim(self, vh, vr, nh, nr, lh, lr):
        delta = float(lh - lr)
        val = [0.0] * self.n
        for k in range(self.n):
            for ngram in vh[k]:
                val[k] += min(vh[k][ngram], vr[k].get(ngram, 0.0)) * vr[k].get(ngram, 0.0)
            if nh[k] != 0 and nr[k] != 0:
                val[k] /= nh[k] * nr[k]
            val[k] *= math.e ** (-(delta ** 2) / (2 * self.sigma ** 2))
        return val

    def compute_score(self):
        """-> (corpus CIDEr, [per-sample CIDEr])."""
        df = defaultdict(float)
        for refs in self.crefs:
            for ngram in set(ng for ref in refs for ng in ref):
                df[ngram] += 1
        ref_len = math.log(float(len(self.crefs)))
        scores = []
        for test, refs in zip(self.ctest, self.crefs):
            vh, nh, lh = self._vec(test, df, ref_len)
            tot = [0.0] * self.n
            for ref in refs:
                vr, nr, lr = self._vec(ref, df, ref_len)
                tot = [a + b for a, b in zip(tot, self._sim(vh, vr, nh, nr, lh, lr))]
            scores.append(sum(tot) / self.n / len(refs) * 10.0)
        return (sum(scores) / len(scores) if scores else 0.0), scores
