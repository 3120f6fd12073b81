"""Lane-level numpy emulation of the ingest kernels' two non-trivial algorithms
(myrrix-recommender_amd/csrc/ingest_kernels.h), so that their logic is checked on the CPU-only box:

  * the stable radix scatter: per-workgroup tile, per-wave multi-split ranking with 8 ballots per
    64 keys and a running per-digit count, digit segments of the tile, LDS staging, run write-out;
  * the per-pair record replay (NaN removes, first value starts, later values are added in fp32).
Design check, not product code and not the oracle."""
import numpy as np

WAVE = 64


def ballot(mask):
    return sum(1 << int(l) for l in np.flatnonzero(mask))


def radix_pass(keys, pay, shift, wave_tile=256):
    """One LSD pass exactly as rs_histogram_kernel + scan + rs_scatter_kernel do it (4 waves per tile)."""
    n = len(keys)
    block_tile = 4 * wave_tile
    n_blocks = (n + block_tile - 1) // block_tile
    digit = ((keys >> np.uint64(shift)) & np.uint64(255)).astype(np.int64)
    counts = np.zeros((256, n_blocks), np.int64)                     # rs_histogram_kernel
    for b in range(n_blocks):
        d = digit[b * block_tile:(b + 1) * block_tile]
        counts[:, b] = np.bincount(d, minlength=256)
    offsets = np.cumsum(np.concatenate([[0], counts.reshape(-1)[:-1]])).reshape(256, n_blocks)  # digit-major scan
    out_k, out_p = np.empty_like(keys), np.empty_like(pay)
    for b in range(n_blocks):
        cnt = np.zeros((4, 256), np.int64)
        lrank = {}
        for w in range(4):                                           # pass 1: rank inside the wave
            base = b * block_tile + w * wave_tile
            for r in range(wave_tile // WAVE):
                idx = base + WAVE * r + np.arange(WAVE)
                ok = idx < n
                dg = np.where(ok, digit[np.minimum(idx, n - 1)], 0)
                for lane in range(WAVE):
                    if not ok[lane]:
                        continue
                    peers = ballot(ok)
                    for bit in range(8):
                        m = ballot(ok & (((dg >> bit) & 1) == 1))    # __ballot(one): inactive lanes vote 0
                        peers &= m if (dg[lane] >> bit) & 1 else ~m
                    lt = (1 << lane) - 1
                    lrank[int(idx[lane])] = cnt[w, dg[lane]] + bin(peers & lt).count("1")
                for dd in np.unique(dg[ok]):                          # the highest peer advances the count
                    cnt[w, dd] += int(np.sum(dg[ok] == dd))
        tot = cnt.sum(axis=0)
        seg = np.concatenate([[0], np.cumsum(tot)[:-1]])             # block_exclusive_scan_256
        woff = seg[None, :] + np.cumsum(np.vstack([np.zeros(256, np.int64), cnt[:-1]]), axis=0)
        lo, hi = b * block_tile, min(n, (b + 1) * block_tile)
        skey = np.zeros(hi - lo, keys.dtype)
        spay = np.zeros(hi - lo, pay.dtype)
        for i in range(lo, hi):                                       # pass 2: into LDS in digit order
            w = (i - lo) // wave_tile
            q = woff[w, digit[i]] + lrank[i]
            skey[q], spay[q] = keys[i], pay[i]
        for q in range(hi - lo):                                      # pass 3: runs out to global
            dd = int((skey[q] >> np.uint64(shift)) & np.uint64(255))
            pos = offsets[dd, b] + (q - seg[dd])
            out_k[pos], out_p[pos] = skey[q], spay[q]
    return out_k, out_p


def radix_sort(keys, pay, wave_tile=256, first_digit=0):
    for d in range(first_digit, 8):
        dg = (keys >> np.uint64(8 * d)) & np.uint64(255)
        if np.all(dg == dg[0]):
            continue                                                  # rs_digit_totals_kernel: trivial digit
        keys, pay = radix_pass(keys, pay, 8 * d, wave_tile)
    return keys, pay


def replay_pair(values, thr):
    """replay_pairs_kernel for one pair: returns (alive, keep, value)."""
    present, v = False, np.float32(0)
    for x in np.asarray(values, np.float32):
        if np.isnan(x):
            present = False
        elif not present:
            present, v = True, x
        else:
            v = np.float32(v + x)
    return present, bool(present and not abs(v) < np.float32(thr)), v
