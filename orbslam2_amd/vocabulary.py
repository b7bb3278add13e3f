"""Descriptor -> word quantisation of DBoW2's ORB vocabulary on the GPU
(TemplatedVocabulary<FORB>::transform, Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1126-1259 of the
reference)."""
import ctypes as C

import numpy as np

from . import _lib as L


class ORBVocabulary:
    """A vocabulary tree in flat form: desc (N,32) uint8, child_start (N+1,), children, word_id (N,),
    weight (N,) float64, depth L.  Node 0 is the root; children come after their parent."""

    def __init__(self, desc, child_start, children, word_id, weight, depth, device=0):
        self._lib = L.lib()
        self._h = C.c_void_p()
        a = [np.ascontiguousarray(desc, np.uint8).reshape(-1, 32), np.ascontiguousarray(child_start, np.int32),
             np.ascontiguousarray(children, np.int32), np.ascontiguousarray(word_id, np.int32),
             np.ascontiguousarray(weight, np.float64)]
        L.check(self._lib.orbx_vocabulary_create(device, *[L.ptr(x) for x in a], len(a[0]), depth, C.byref(self._h)),
                "orbx_vocabulary_create")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.orbx_vocabulary_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def transform_features(self, descriptors, levelsup=4):
        """Per descriptor: (word id, node id at `levelsup` levels above the leaves, word weight)."""
        d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        n = len(d)
        word = np.zeros(max(n, 1), np.int32)
        node = np.zeros(max(n, 1), np.int32)
        wt = np.zeros(max(n, 1), np.float64)
        L.check(self._lib.orbx_vocabulary_transform(self._h, L.ptr(d), n, levelsup, L.ptr(word), L.ptr(node),
                                                    L.ptr(wt)), "orbx_vocabulary_transform")
        return word[:n], node[:n], wt[:n]

    def transform_features_device(self, d_descriptors, n, d_word, d_node, d_weight, levelsup=4, stream=0):
        """The same for n descriptors in device memory; all pointers are device addresses (ints), outputs
        int32 / int32 / float64 arrays of n entries.  Asynchronous on `stream` (a hipStream_t, 0 = default)."""
        P = C.c_void_p
        L.check(self._lib.orbx_vocabulary_transform_device(self._h, P(stream), P(d_descriptors), n, levelsup, P(d_word),
                                                           P(d_node), P(d_weight)), "orbx_vocabulary_transform_device")

    def transform(self, descriptors, levelsup=4):
        """Frame::ComputeBoW (src/Frame.cc:422-431): (BowVector as {word: value}, FeatureVector as
        {node: [feature indices]}) for TF_IDF weighting and L1 scoring, the configuration of ORBvoc
        (TemplatedVocabulary.h:1145-1193: weights add up per word in feature order, then the vector is
        divided by the sum of absolute values)."""
        word, node, wt = self.transform_features(descriptors, levelsup)
        bow, fv = {}, {}
        for i in range(len(word)):
            if wt[i] > 0:
                bow[int(word[i])] = bow.get(int(word[i]), 0.0) + float(wt[i])
                fv.setdefault(int(node[i]), []).append(i)
        norm = sum(abs(v) for _, v in sorted(bow.items()))
        if norm > 0:
            bow = {k: v / norm for k, v in bow.items()}
        return bow, fv
