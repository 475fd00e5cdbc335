"""RANSAC subsets drawn from Python's RNG stream exactly as the reference draws them, in bulk.

The reference draws every minimal sample with ``random.sample(range(n), k)`` on the GLOBAL generator
(campose_processor.py:531, epipolar_processor.py:225): the drop-in has to consume that stream the same way, or every
later draw of the pipeline differs.  ``[random.sample(range(n), k) for _ in range(count)]`` costs ~2 us per index in
the interpreter -- 0.6-1.2 ms for the 300 six-point samples of a view, as much as the device spends on the whole
RANSAC.  ``sample_indices`` returns the same lists and leaves the generator in the same state, but takes the
generator's 32-bit words in one ``getrandbits`` call and replays CPython's algorithm on them with NumPy:

* ``sample`` of a range larger than ``setsize`` (85 for k = 6 ... 21) picks ``k`` distinct ``_randbelow(n)`` values,
  re-drawing duplicates (random.py ``Random.sample``, set branch);
* ``_randbelow(n)`` takes ``getrandbits(n.bit_length())`` until the value is below ``n``; ``getrandbits(b)``, b <= 32,
  is the next 32-bit word of the Mersenne Twister shifted right by ``32 - b``; ``getrandbits(32 m)`` is the next m words,
  least significant first.

Everything outside that envelope (small populations: the pool branch; populations above 2^32; an interpreter whose
``random`` behaves differently -- checked once against ``random.sample`` itself on a scratch generator) falls back to the
plain list comprehension.
"""
import random
from math import ceil, log

import numpy as np

_verified = None


def _plain(rng, n, k, count):
    return [rng.sample(range(n), k) for _ in range(count)]


def _setsize(k):
    size = 21
    if k > 5:
        size += 4 ** ceil(log(k * 3, 4))
    return size


def _bulk(rng, n, k, count):
    bits = n.bit_length()
    need = count * k
    state = rng.getstate()
    accept = n / float(1 << bits)
    m = int(need / accept * 1.15) + 256                       # words drawn; what is not consumed is given back below
    words = np.frombuffer(rng.getrandbits(32 * m).to_bytes(4 * m, "little"), dtype="<u4")
    cand = words >> np.uint32(32 - bits)
    pos = np.flatnonzero(cand < n)                             # words _randbelow accepts
    vals = cand[pos].astype(np.int64)
    # k accepted words per sample, except where a sample meets a value it already holds and re-draws: rows are taken in
    # blocks up to the first such sample, that one is replayed word by word, and the blocking restarts behind it
    out, off, total = [], 0, int(vals.shape[0])
    while len(out) < count:
        take = min(count - len(out), (total - off) // k)
        first_dup = take
        if take > 0:
            block = vals[off:off + take * k].reshape(take, k)
            srt = np.sort(block, axis=1)
            dup_rows = np.flatnonzero(np.any(srt[:, 1:] == srt[:, :-1], axis=1))
            if dup_rows.shape[0]:
                first_dup = int(dup_rows[0])
            out.extend(block[:first_dup].tolist())
            off += first_dup * k
        if len(out) < count:                                   # the sample at `off` re-draws (or the drawn words run out)
            picked, seen = [], set()
            while len(picked) < k and off < total:
                j = int(vals[off]); off += 1
                if j not in seen:
                    seen.add(j); picked.append(j)
            if len(picked) < k:                                # ran out of drawn words (never seen; 15 % + 256 spare)
                rng.setstate(state)
                return _plain(rng, n, k, count)
            out.append(picked)
    used = int(pos[off - 1]) + 1
    rng.setstate(state)
    rng.getrandbits(32 * used)                                 # consume exactly the words random.sample would have
    return out


def _self_check():
    global _verified
    if _verified is None:
        ok = True
        for n, k, count, seed in ((5000, 6, 40, 1), (97, 8, 50, 2), (1 << 20, 6, 30, 3), (130, 6, 200, 4)):
            a, b = random.Random(seed), random.Random(seed)
            ok = ok and _bulk(a, n, k, count) == _plain(b, n, k, count) and a.getstate() == b.getstate()
        _verified = ok
    return _verified


def sample_indices(n, k, count, rng=None, as_array=False):
    """``[random.sample(range(n), k) for _ in range(count)]`` -- same lists, same generator state afterwards.
    ``rng``: a ``random.Random`` (default: the module-level generator the reference uses); ``as_array``: return the
    samples as an int32 array (count, k) instead of a list of lists."""
    n, k, count = int(n), int(k), int(count)
    if rng is None:
        rng = getattr(random, "_inst", None)          # the hidden generator behind random.sample / random.seed (CPython)
        if rng is None:                               # another implementation of the module: its own sample, one by one
            out = [random.sample(range(n), k) for _ in range(count)]
            return np.asarray(out, dtype=np.int32).reshape(count, k) if as_array else out
    # below ~400 items so many samples re-draw a duplicate (15 / n of them for k = 6) that the replay is no faster
    if (count * k < 64 or n < max(400, _setsize(k) + 1) or n.bit_length() > 32 or type(rng).getrandbits is not random.Random.getrandbits
            or not _self_check()):
        out = _plain(rng, n, k, count)
    else:
        out = _bulk(rng, n, k, count)
    return np.asarray(out, dtype=np.int32).reshape(count, k) if as_array else out
