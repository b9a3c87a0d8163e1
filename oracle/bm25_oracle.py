"""BM25 + tokeniser restatement — TEST INFRASTRUCTURE ONLY (see oracle/oracle.h).

Follows src/index/bm25.rs line by line in numpy float32 scalars (pure-Python loops: small cases only):
    K1, B            bm25.rs:9-10
    tokenize         bm25.rs:127-132   regex [a-zA-Z0-9]+, lowercase, drop 1-char tokens
    Bm25Scorer.build bm25.rs:33-74
    score_query      bm25.rs:77-106    idf :88, norm :97, score :100
    search           bm25.rs:109-122   positives only, stable sort descending, truncate
hybrid_rerank (bm25.rs:135-170) lives in oracle.c (orc_hybrid_rerank).
"""
import ctypes
import ctypes.util
import re

import numpy as np

f32 = np.float32
K1 = f32(1.2)
B = f32(0.75)
_TOKEN = re.compile(r"[a-zA-Z0-9]+")
# f32::ln is the platform libm's logf (what the C++ host's std::log(float) calls too): bit-identical, unlike log(double) rounded
_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.logf.restype = ctypes.c_float
_libm.logf.argtypes = [ctypes.c_float]


def logf(x):
    return f32(_libm.logf(float(x)))


def tokenize(text):
    return [m.group(0).lower() for m in _TOKEN.finditer(text) if len(m.group(0)) > 1]


class Bm25Scorer:
    def __init__(self, doc_freq, num_docs, avg_doc_len, doc_lengths, term_freqs):
        self.doc_freq, self.num_docs, self.avg_doc_len = doc_freq, num_docs, avg_doc_len
        self.doc_lengths, self.term_freqs = doc_lengths, term_freqs

    @classmethod
    def build(cls, documents):
        doc_freq, doc_lengths, term_freqs, total = {}, [], [], 0
        for doc in documents:
            tokens = tokenize(doc)
            doc_lengths.append(len(tokens))
            total += len(tokens)
            tf, seen = {}, set()
            for t in tokens:
                tf[t] = tf.get(t, 0) + 1
                if t not in seen:
                    doc_freq[t] = doc_freq.get(t, 0) + 1
                    seen.add(t)
            term_freqs.append(tf)
        n = len(documents)
        avg = f32(total) / f32(n) if n > 0 else f32(1.0)
        return cls(doc_freq, n, avg, doc_lengths, term_freqs)

    def score_query(self, query):
        scores = np.zeros(self.num_docs, f32)
        for token in tokenize(query):
            df = f32(self.doc_freq.get(token, 0))
            if df == 0:
                continue
            ratio = f32(f32(f32(self.num_docs) - df) + f32(0.5)) / f32(df + f32(0.5))
            idf = logf(f32(ratio + f32(1.0)))
            for doc_id, tfm in enumerate(self.term_freqs):
                tf = f32(tfm.get(token, 0))
                if tf == 0:
                    continue
                doc_len = f32(self.doc_lengths[doc_id])
                norm = f32(f32(f32(1.0) - B) + f32(B * f32(doc_len / self.avg_doc_len)))
                score = f32(f32(idf * f32(tf * f32(K1 + f32(1.0)))) / f32(tf + f32(K1 * norm)))
                scores[doc_id] = f32(scores[doc_id] + score)
        return scores

    def search(self, query, top_k):
        scores = self.score_query(query)
        scored = [(i, float(s)) for i, s in enumerate(scores) if s > 0]
        scored.sort(key=lambda t: -t[1])  # stable, like sort_by
        return scored[:top_k]
