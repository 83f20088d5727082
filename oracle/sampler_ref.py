"""TEST INFRASTRUCTURE (oracle): CPU restatement of the GPU-resident episode sampler (fumi_amd/csrc/sampler.hip).

What it restates: the episode construction of fumi/dataset/data.py:294-581 + torchmeta (CombinationMetaDataset: N distinct
classes per task; ClassSplitter(shuffle=True): a random K / Q split of each class's samples; ConcatTask: class-major order,
categorical labels 0..N-1).  torchmeta's own RNG streams cannot be reproduced (the package is not in /root/reference), so
the *semantics* are the contract and the random stream is this project's counter-based hash: the device kernel must agree
with this file bit for bit (integer work)."""
import numpy as np

M32 = 0xFFFFFFFF


def mix(x):
    x &= M32
    x ^= x >> 16; x = (x * 0x7feb352d) & M32
    x ^= x >> 15; x = (x * 0x846ca68b) & M32
    x ^= x >> 16
    return x


def step_key(seed, step):
    k = mix(seed & M32)
    k = mix(k ^ ((seed >> 32) & M32))
    k = mix(k ^ (step & M32))
    k = mix(k ^ ((step >> 32) & M32))
    return k


def rand_below(key, a, b, c, n):
    r = mix(mix(mix(key ^ ((a * 0x9E3779B9) & M32)) ^ ((b * 0x85EBCA6B) & M32)) ^ ((c * 0xC2B2AE35) & M32))
    return (r * n) >> 32


def sample_distinct(key, a, b, n, m):
    """Floyd's subset algorithm, then a Fisher-Yates shuffle (sampler.hip: sample_distinct)."""
    sel = []
    for j in range(n - m, n):
        t = rand_below(key, a, b, 2 * j, j + 1)
        sel.append(j if t in sel else t)
    for i in range(m - 1, 0, -1):
        t = rand_below(key, a, b, 2 * i + 1, i + 1)
        sel[i], sel[t] = sel[t], sel[i]
    return sel


def sample_episodes(seed, step, B, N, K, Q, class_ptr, class_items):
    """-> classes [B,N], items_s [B,N,K], items_q [B,N,Q] (int64)."""
    C = len(class_ptr) - 1
    key = step_key(int(seed), int(step))
    m = K + Q
    cls = np.zeros((B, N), np.int64); it_s = np.zeros((B, N, K), np.int64); it_q = np.zeros((B, N, Q), np.int64)
    for b in range(B):
        cs = sample_distinct(key, b, 0xFFFF, C, N)
        for n, c in enumerate(cs):
            p0, n_c = int(class_ptr[c]), int(class_ptr[c + 1] - class_ptr[c])
            sel = sample_distinct(key, b, n, n_c, m) if n_c >= m else [i % max(n_c, 1) for i in range(m)]
            cls[b, n] = c
            it_s[b, n] = [class_items[p0 + s] for s in sel[:K]]
            it_q[b, n] = [class_items[p0 + s] for s in sel[K:]]
    return cls, it_s, it_q


def sample_episodes_tm(seed, step, B, N, K, Q, class_ptr, class_items, fixed_split=True):
    """torchmeta's task semantics on the same stream (sampler.hip: sample_episodes_tm_kernel): -> classes, labels (a random
    permutation of 0..N-1 per task: Categorical), items_s, items_q; fixed_split: the members drawn for a class depend only on
    (seed, the task's class tuple) -- ClassSplitter's hash(task) + seed seeding."""
    C = len(class_ptr) - 1
    seed, step = int(seed), int(step)
    skey = mix(mix(seed & M32) ^ ((seed >> 32) & M32))
    key = mix(mix(skey ^ (step & M32)) ^ ((step >> 32) & M32))
    seed_key = mix(skey ^ 0x5bd1e995)
    m = K + Q
    cls = np.zeros((B, N), np.int64); lab = np.zeros((B, N), np.int64)
    it_s = np.zeros((B, N, K), np.int64); it_q = np.zeros((B, N, Q), np.int64)
    for b in range(B):
        cs = sample_distinct(key, b, 0xFFFF, C, N)
        lab[b] = sample_distinct(key, b, 0xFFFE, N, N)
        tk = seed_key
        for n, c in enumerate(cs):
            tk = mix(tk ^ ((c * 0x9E3779B9 + n) & M32))
        for n, c in enumerate(cs):
            p0, n_c = int(class_ptr[c]), int(class_ptr[c + 1] - class_ptr[c])
            if n_c < m:
                sel = [i % max(n_c, 1) for i in range(m)]
            else:
                sel = sample_distinct(tk, 0, n, n_c, m) if fixed_split else sample_distinct(key, b, n, n_c, m)
            cls[b, n] = c
            it_s[b, n] = [class_items[p0 + s_] for s_ in sel[:K]]
            it_q[b, n] = [class_items[p0 + s_] for s_ in sel[K:]]
    return cls, lab, it_s, it_q
