#!/usr/bin/env python3
"""Decode-table size statistics of a packed fixture (tools only; uses the oracle's code log).

For every stream: the prefix codes its meta-block headers build (alphabet size, count per length),
and what a two-level LUT costs under a given root width policy.
usage: tools/table_stats.py [fixture name, default cfg2_streams]
"""
import ctypes
import sys
import os
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import helpers as H


def lut_entries(count, root):
    """entries of a two-level table: 2^root + sub-tables sized like build_table()"""
    maxlen = max([l for l in range(16) if count[l]] or [0])
    if maxlen <= root:
        return 1 << min(root, max(maxlen, 0)) if False else 1 << root, 0
    left = list(count)
    total = 1 << root
    # walk canonical order; open a sub-table per distinct root prefix
    code = 0
    subs = 0
    key_prev = None
    cur_room = 0
    for l in range(1, 16):
        code <<= 1
        for _ in range(count[l]):
            if l > root:
                key = code >> (l - root)
                if key != key_prev:
                    key_prev = key
                    l2 = l
                    room = 1 << (l2 - root)
                    while l2 < 15:
                        room -= left[l2]
                        if room <= 0:
                            break
                        l2 += 1
                        room <<= 1
                    total += 1 << (l2 - root)
                    subs += 1
            left[l] -= 1
            code += 1
    return total, subs


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_streams"
    meta, streams = H.packed_streams(name)
    L = H.oracle()
    cap = 4096
    log = (ctypes.c_uint16 * (17 * cap))()
    ctypes.c_void_p.in_dll(L, "oracle_code_log").value = ctypes.addressof(log)
    ctypes.c_size_t.in_dll(L, "oracle_code_log_cap").value = cap
    rows = []
    for s in streams:
        ctypes.c_size_t.in_dll(L, "oracle_code_log_n").value = 0
        rc, got, used, out = H.oracle_yeast(s)
        n = ctypes.c_size_t.in_dll(L, "oracle_code_log_n").value
        codes = np.frombuffer(log, dtype=np.uint16, count=17 * n).reshape(n, 17).copy()
        rows.append(codes)
    for policy in ("root8", "root=min(8,maxlen)", "root7", "root=min(7,maxlen)", "root6", "root=min(6,maxlen)", "root9",
                   "lit:min(8,maxlen) other:min(9,maxlen)", "canonical"):
        sizes = []
        for codes in rows:
            b = 0
            for c in codes:
                num, count = int(c[0]), [int(x) for x in c[1:]]
                if num == 18 + 0 and False:
                    continue
                maxlen = max([l for l in range(16) if count[l]] or [0])
                nsym = sum(count)
                if policy == "canonical":
                    b += 64 + nsym * (1 if num <= 256 else 2)
                    continue
                if policy.startswith("lit:"):
                    root = min(8 if num == 256 else 9, max(maxlen, 1))
                elif "min" in policy:
                    root = min(int(policy[9]), max(maxlen, 1))
                else:
                    root = int(policy[4:])
                e, _ = lut_entries(count, root)
                b += 2 * e
            sizes.append(b)
        a = np.array(sizes)
        print("%-40s bytes/stream: mean %6.0f  p50 %6.0f  p90 %6.0f  p99 %6.0f  max %6.0f" %
              (policy, a.mean(), np.percentile(a, 50), np.percentile(a, 90), np.percentile(a, 99), a.max()))
    ncodes = np.array([len(c) for c in rows])
    print("codes per stream (incl. context-map codes): mean %.1f max %d" % (ncodes.mean(), ncodes.max()))
    allc = np.concatenate(rows)
    for num in sorted(set(allc[:, 0])):
        sel = allc[allc[:, 0] == num]
        maxlen = np.array([max([l for l in range(16) if r[1 + l]] or [0]) for r in sel])
        nsym = sel[:, 1:].sum(1)
        print("alphabet %4d: %5d codes, symbols mean %.0f max %d, max length mean %.1f p90 %d max %d" %
              (num, len(sel), nsym.mean(), nsym.max(), maxlen.mean(), np.percentile(maxlen, 90), maxlen.max()))


if __name__ == "__main__":
    main()


def second_level_share():
    """share of lookups that need a second level, by alphabet and root width, if symbols occur
    with probability 2^-length"""
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_streams"
    meta, streams = H.packed_streams(name)
    L = H.oracle()
    cap = 4096
    log = (ctypes.c_uint16 * (17 * cap))()
    ctypes.c_void_p.in_dll(L, "oracle_code_log").value = ctypes.addressof(log)
    ctypes.c_size_t.in_dll(L, "oracle_code_log_cap").value = cap
    acc = {}
    for s in streams:
        ctypes.c_size_t.in_dll(L, "oracle_code_log_n").value = 0
        H.oracle_yeast(s)
        n = ctypes.c_size_t.in_dll(L, "oracle_code_log_n").value
        codes = np.frombuffer(log, dtype=np.uint16, count=17 * n).reshape(n, 17)
        for c in codes:
            num = int(c[0])
            kind = "literal" if num == 256 else "iac" if num == 704 else "dist" if 64 <= num <= 520 else None
            if kind is None:
                continue
            cnt = c[1:].astype(np.float64)
            for R in (4, 5, 6, 7, 8):
                p = sum(cnt[l] * 2.0 ** -l for l in range(R + 1, 16))
                acc.setdefault((kind, R), []).append(p)
    for k in sorted(acc):
        print("%-8s root %d: second-level share mean %.3f  max %.3f" % (k[0], k[1], np.mean(acc[k]), np.max(acc[k])))


if __name__ == "__main__" and os.environ.get("SECOND"):
    second_level_share()
