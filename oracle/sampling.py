"""TEST INFRASTRUCTURE -- CPU restatement of the on-device sampling rule (mi_forward_tokens).

The reference delegates on-device sampling to NxDI (absent from /root/reference); what it fixes is
the contract: per-request rows (top_k, top_p, temperature), greedy requests rewritten to top_k = 1
(vllm_neuron/worker/neuronx_distributed_model_runner.py:1106-1140), and ids returned in place of
logits (neuronx_distributed_model_loader.py:352-356, 367-375).  Parity unpinned beyond that: no
reference test holds a sampled-id vector.  This file DEFINES the rule the HIP kernel implements:

  top_k == 1 : argmax, lowest index on ties.
  otherwise  : candidates = the top_k logits ordered (value desc, index asc);
               p_i = exp((l_i - l_max) / temperature)   (fp32, accumulated serially in that order);
               nucleus: keep candidate i while the mass before i is < top_p * total (first always kept);
               u = (splitmix64(seed ^ splitmix64(0x5EED + row)) >> 40) / 2**24;
               pick the first i whose cumulative mass exceeds u * kept.
"""
import numpy as np

MASK = (1 << 64) - 1
MAX_TOP_K = 256


def splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & MASK
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & MASK
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & MASK
    return x ^ (x >> 31)


def uniform(seed: int, row: int) -> np.float32:
    r = splitmix64((seed & MASK) ^ splitmix64((0x5EED + row) & MASK))
    return np.float32(r >> 40) * np.float32(1.0 / 16777216.0)


def candidates(logits: np.ndarray, top_k: int):
    """(indices, values) of the top_k logits, value descending, index ascending on ties."""
    top_k = max(1, min(int(top_k), MAX_TOP_K, logits.shape[0]))
    order = np.lexsort((np.arange(logits.shape[0]), -logits.astype(np.float64)))[:top_k]
    return order, logits[order].astype(np.float32)


def kept_distribution(logits: np.ndarray, top_k, top_p, temperature):
    """(indices, fp32 masses) of the nucleus the sampler draws from."""
    idx, val = candidates(logits, top_k)
    w = np.exp((val - val[0]) * (np.float32(1.0) / np.float32(temperature)), dtype=np.float32)
    total = np.float32(0)
    for x in w:
        total = np.float32(total + x)
    limit = np.float32(np.float32(top_p) * total)
    kept, n = np.float32(0), 0
    for i, x in enumerate(w):
        if i > 0 and not kept < limit:
            break
        kept = np.float32(kept + x)
        n = i + 1
    return idx[:n], w[:n]


def sample_row(logits: np.ndarray, top_k, top_p, temperature, seed: int, row: int) -> int:
    logits = np.asarray(logits, dtype=np.float32)
    if int(top_k) <= 1:
        return int(np.argmax(logits))            # numpy: first maximum
    idx, w = kept_distribution(logits, top_k, top_p, temperature)
    kept = np.float32(0)
    for x in w:
        kept = np.float32(kept + x)
    target = np.float32(uniform(seed, row) * kept)
    cum = np.float32(0)
    for i, x in enumerate(w):
        cum = np.float32(cum + x)
        if cum > target:
            return int(idx[i])
    return int(idx[-1])
