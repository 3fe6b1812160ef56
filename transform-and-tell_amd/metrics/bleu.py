"""BLEU-1..4 as the reference's evaluate-mode bookkeeping computes it (transformer_faces_objects.py:109-116:
`BleuScorer(n=4)`, `+= (generation, [caption])`, `compute_score(option='closest')`).  The scorer is pycocoevalcap's
(third-party, absent here - a host-side text metric, not part of the GPU path); this is a restatement of its published
algorithm: clipped n-gram precision against the references' maximum counts, geometric mean over orders 1..k, brevity
penalty exp(1 - 1/ratio) when the hypothesis is shorter than the closest reference length, with the library's
`tiny = 1e-15` / `small = 1e-9` smoothing constants."""
import math
from collections import defaultdict


def precook(s, n=4):
    words = s.split()
    counts = defaultdict(int)
    for k in range(1, n + 1):
        for i in range(len(words) - k + 1):
            counts[tuple(words[i:i + k])] += 1
    return len(words), counts


def cook_refs(refs, n=4):
    reflen, maxcounts = [], {}
    for ref in refs:
        rl, counts = precook(ref, n)
        reflen.append(rl)
        for ngram, count in counts.items():
            maxcounts[ngram] = max(maxcounts.get(ngram, 0), count)
    return reflen, maxcounts


def cook_test(test, refs, n=4):
    reflen, refmaxcounts = refs
    testlen, counts = precook(test, n)
    result = {'testlen': testlen, 'reflen': reflen, 'guess': [max(0, testlen - k + 1) for k in range(1, n + 1)],
              'correct': [0] * n}
    for ngram, count in counts.items():
        result['correct'][len(ngram) - 1] += min(refmaxcounts.get(ngram, 0), count)
    return result


class BleuScorer:
    def __init__(self, test=None, refs=None, n=4):
        self.n = n
        self.crefs, self.ctest = [], []
        if refs is not None:
            self += (test, refs)

    def __iadd__(self, other):
        test, refs = other
        cooked = cook_refs(refs, self.n)
        self.crefs.append(cooked)
        self.ctest.append(cook_test(test, cooked, self.n) if test is not None else None)
        return self

    @staticmethod
    def _single_reflen(reflens, option, testlen):
        if option == 'shortest':
            return min(reflens)
        if option == 'average':
            return float(sum(reflens)) / len(reflens)
        if option == 'closest':
            return min((abs(l - testlen), l) for l in reflens)[1]
        raise ValueError('unsupported reflen option %s' % option)

    def compute_score(self, option='closest', verbose=0):
        """-> ([BLEU-1 .. BLEU-n] of the corpus, per-segment lists)."""
        n, small, tiny = self.n, 1e-9, 1e-15
        bleu_list = [[] for _ in range(n)]
        tot = {'testlen': 0, 'reflen': 0, 'guess': [0] * n, 'correct': [0] * n}
        for comps in self.ctest:
            testlen = comps['testlen']
            reflen = self._single_reflen(comps['reflen'], option, testlen)
            tot['testlen'] += testlen
            tot['reflen'] += reflen
            for key in ('guess', 'correct'):
                for k in range(n):
                    tot[key][k] += comps[key][k]
            bleu = 1.0
            for k in range(n):
                bleu *= (float(comps['correct'][k]) + tiny) / (float(comps['guess'][k]) + small)
                bleu_list[k].append(bleu ** (1.0 / (k + 1)))
            ratio = (testlen + tiny) / (reflen + small)
            if ratio < 1:
                for k in range(n):
                    bleu_list[k][-1] *= math.exp(1 - 1 / ratio)
        bleus, bleu = [], 1.0
        for k in range(n):
            bleu *= float(tot['correct'][k] + tiny) / (tot['guess'][k] + small)
            bleus.append(bleu ** (1.0 / (k + 1)))
        ratio = (tot['testlen'] + tiny) / (tot['reflen'] + small)
        if ratio < 1:
            for k in range(n):
                bleus[k] *= math.exp(1 - 1 / ratio)
        return bleus, bleu_list
