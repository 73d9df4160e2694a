#!/usr/bin/env python3
"""tools/isa_hist.py -- instruction histogram of the batch kernel's command loop, from the compiler's own listing.

    python3 tools/isa_hist.py [function-substring ...]      (default: the third loop form in its two settings and round 2's ring form)

Compiles brotli_amd/csrc/yeast_lane.hip to gfx950 assembly with the product's flags (hipcc -S, device only; no GPU
needed), finds in each named function its largest loop (label ... backward branch to that label) and counts the
instructions between the two by class.  Counted apart as "rare blocks": the blocks the loop jumps to that lie behind its back edge (marked unlikely in the
source) and the spans inside it that a wavefront-uniform branch skips and that hold a loop of their own (block
switches): a trip that meets none of the rare cases runs neither.  Static
counts: what the listing holds, not what a trip executes -- inner loops (table reloads after a block switch) and
blocks skipped by an empty exec mask are in the sum.  VERDICT r02, item 1 asked for this table before and after.
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ("-O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -enable-ipra=false "
         "-mllvm -amdgpu-sched-strategy=max-ilp -S --cuda-device-only").split()


def listing():
    out = "/tmp/yeast_lane_hist.s"
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS +
                          ["yeast_lane.hip", "-o", out], cwd=os.path.join(ROOT, "brotli_amd", "csrc"),
                          stderr=subprocess.DEVNULL)
    return open(out).read().splitlines()


def klass(op):
    if op.startswith("v_cndmask"):
        return "v_cndmask"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "v_readlane/readfirstlane"
    if op.startswith("v_cmp"):
        return "v_cmp"
    if op.startswith("v_"):
        return "other vector ALU"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vector memory"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith(("s_load", "s_buffer_load")):
        return "scalar memory"
    if op.startswith("s_"):
        return "scalar ALU"
    return "other"


def functions(lines):
    out, name, start = {}, None, 0
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):\s*; @", ln)
        if m:
            name, start = m.group(1), i
        elif name and ln.startswith(".Lfunc_end"):
            out[name] = (start, i)
            name = None
    return out


def instr(ln):
    s = ln.strip()
    if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
        return None
    return s.split()[0]


def analyse(lines, lo, hi):
    label_at = {}
    for i in range(lo, hi):
        m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
        if m:
            label_at[m.group(1)] = i
    best = None
    for i in range(lo, hi):
        op = instr(lines[i])
        if op and op.startswith(("s_cbranch", "s_branch")):
            tgt = lines[i].split()[-1]
            if tgt in label_at and label_at[tgt] < i and (best is None or i - label_at[tgt] > best[1] - best[0]):
                best = (label_at[tgt], i)
    a, b = best
    # spans inside the loop that a wavefront-uniform forward branch skips and that hold a loop of their own (the table
    # reloads of a block switch): not on the path of a trip without a block switch
    skip = set()
    for i in range(a, b + 1):
        op = instr(lines[i])
        if op and op.startswith(("s_cbranch_vcc", "s_cbranch_scc")):
            tgt = label_at.get(lines[i].split()[-1], -1)
            if i < tgt <= b and tgt - i < 0.45 * (b - a):
                inner = False
                for j in range(i + 1, tgt):
                    oj = instr(lines[j])
                    if oj and oj.startswith("s_cbranch"):
                        tj = label_at.get(lines[j].split()[-1], -1)
                        if i < tj < j:
                            inner = True
                if inner:
                    skip.update(range(i + 1, tgt))
    body = collections.Counter(klass(op) for op in (instr(lines[i]) for i in range(a, b + 1) if i not in skip) if op)
    inl = collections.Counter(klass(op) for op in (instr(lines[i]) for i in sorted(skip)) if op)
    # out-of-line blocks: labels behind the back edge that the body branches to; each runs to its s_branch back
    rare = collections.Counter()
    targets = set()
    for i in range(a, b + 1):
        op = instr(lines[i])
        if op and op.startswith("s_cbranch"):
            tgt = lines[i].split()[-1]
            if tgt in label_at and label_at[tgt] > b:
                targets.add(label_at[tgt])
    for t in sorted(targets):
        for i in range(t + 1, hi):
            op = instr(lines[i])
            if op:
                rare[klass(op)] += 1
                if op.startswith("s_branch"):
                    break
    rare.update(inl)
    return body, rare, (a - lo, b - lo)


def main():
    want = sys.argv[1:] or ["run_trip3_fnILi32ELj128E", "run_trip3_fnILi32ELj64E", "run_trip_fnILi544E"]
    titles = {"run_trip3_fnILi32ELj128E": "run_trip3_fn<32, 128, 64>: the third loop form with 64-byte output chunks (round 3, the bench's kernel)",
              "run_trip3_fnILi32ELj64E": "run_trip3_fn<32, 64, 16>: the third loop form with 16-byte chunks (batches that do not fill the GPU)",
              "run_trip_fnILi544E": "run_trip_fn<32 | ring>: round 2's write-behind loop (YEAST_MERGE=2)"}
    lines = listing()
    fns = functions(lines)
    order = ["other vector ALU", "v_cndmask", "v_cmp", "v_readlane/readfirstlane", "scalar ALU", "scalar memory", "branch",
             "LDS", "vector memory", "s_waitcnt", "s_nop", "other"]
    for w in want:
        for name, (lo, hi) in fns.items():
            if w in name:
                body, rare, span = analyse(lines, lo, hi)
                print("%s\n  %s, loop at lines %d..%d of the function" % (titles.get(w, w), name, span[0], span[1]))
                print("  %-28s %8s %22s" % ("class", "loop", "rare blocks"))
                for k in order:
                    if body[k] or rare[k]:
                        print("  %-28s %8d %22d" % (k, body[k], rare[k]))
                vec = sum(body[k] for k in order[:4])
                sca = body["scalar ALU"] + body["scalar memory"] + body["branch"] + body["s_waitcnt"] + body["s_nop"]
                print("  %-28s %8d %22d" % ("all", sum(body.values()), sum(rare.values())))
                print("  vector ALU %d, scalar side (ALU, memory, branches, waits, nops) %d, LDS %d, vector memory %d\n"
                      % (vec, sca, body["LDS"], body["vector memory"]))


if __name__ == "__main__":
    main()
