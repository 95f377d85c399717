"""Dynamic census of the on-chip interior point kernel's iteration loop from the compiled gfx950 assembly: the loops of
ipm_onchip_kernel<2,10,0,128> are found from their backward branches, the stage loop of the backward sweep (the innermost loop
that holds the Schur-complement MFMA and the permlane swaps of the elimination) is classified instruction by instruction, and an
iteration is priced as (iteration loop body outside the stage loop) + (N - 1) x (stage loop body).
    python tools/isa_loops.py [out.txt]        (compiles csrc/miqp_gpu.hip to assembly first; ISA_REUSE=1: reuse /tmp/isa_report/k.s)"""
import collections, os, re, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "..", "planner_miqp_amd", "csrc", "miqp_gpu.hip")
SYM = "_ZN4miqp17ipm_onchip_kernelILi2ELi10ELi0ELi128EEEvNS_6DevBufE"


def cls(t):
    op = t.split()[0]
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_")): return "vmem"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_"): return "salu"
    if op.startswith(("v_readlane", "v_readfirstlane")): return "v_readlane (SGPR spill reloads, uniform values)"
    if op.startswith("v_writelane"): return "v_writelane (SGPR spills)"
    if "dpp" in t or "row_" in t: return "dpp moves"
    if op.startswith("v_permlane"): return "permlane swaps"
    if re.match(r"v_(fma|mul|add|max|min|rcp|rsq|div|sub|fmac)_f64", op): return "f64 arithmetic"
    if op.startswith("v_cndmask"): return "selects (v_cndmask)"
    if op.startswith(("v_mov", "v_accvgpr")): return "moves"
    if op.startswith("v_cmp"): return "compares"
    if op.startswith("v_"): return "integer / address VALU"
    return "other"


VALU = {"f64 arithmetic", "moves", "selects (v_cndmask)", "dpp moves", "permlane swaps", "compares", "integer / address VALU",
        "v_readlane (SGPR spill reloads, uniform values)", "v_writelane (SGPR spills)"}


def main():
    tmp = "/tmp/isa_report"; os.makedirs(tmp, exist_ok=True)
    if not os.environ.get("ISA_REUSE"):
        flags = ["--offload-arch=gfx950", "-O3", "-fno-math-errno", "-freciprocal-math", "-fno-signed-zeros", "-fno-trapping-math", "-std=c++17"]
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["--cuda-device-only", "-S", "-o", tmp + "/k.s", SRC], stderr=subprocess.DEVNULL)
    txt = open(tmp + "/k.s").read()
    a = txt.index("\n" + SYM + ":"); b = txt.index(".Lfunc_end", a)
    labels = {}; ins = []
    for ln in txt[a:b].split("\n"):
        t = ln.split(";")[0].strip()
        if not t: continue
        m = re.match(r"(\.LBB[0-9_]+):", t)
        if m: labels[m.group(1)] = len(ins); continue
        if t.startswith(".") or t.endswith(":"): continue
        ins.append(t)
    loops = []
    for i, t in enumerate(ins):
        m = re.match(r"s_cbranch_\w+\s+(\.LBB[0-9_]+)|s_branch\s+(\.LBB[0-9_]+)", t)
        if m:
            lb = m.group(1) or m.group(2)
            if lb in labels and labels[lb] <= i: loops.append((labels[lb], i))
    # stage loop: the smallest loop with >= 1 MFMA and >= 8 permlane swaps (the elimination of the input block)
    cand = [(e - s, s, e) for s, e in loops if sum(1 for t in ins[s:e + 1] if t.startswith("v_mfma")) >= 1 and sum(1 for t in ins[s:e + 1] if t.startswith("v_permlane")) >= 8]
    _, ss, se = min(cand)
    # iteration loop: the smallest loop that contains the stage loop and at least 2000 instructions more
    outer = [(e - s, s, e) for s, e in loops if s < ss and e > se and (e - s) - (se - ss) > 2000]
    _, os_, oe = min(outer)
    stage = collections.Counter(cls(t) for t in ins[ss:se + 1])
    rest = collections.Counter(cls(t) for t in ins[os_:ss] + ins[se + 1:oe + 1])
    N = 20
    out = ["ipm_onchip_kernel<2,10,0,128>: census of the compiled iteration loop (static counts per loop body; an iteration of a 20-step node = rest + 19 x stage)",
           "%-52s %8s %8s %12s" % ("class", "stage", "rest", "iteration")]
    tot = collections.Counter()
    for k in sorted(set(stage) | set(rest), key=lambda k: -(rest[k] + (N - 1) * stage[k])):
        it = rest[k] + (N - 1) * stage[k]; tot[k] = it
        out.append("%-52s %8d %8d %12d" % (k, stage[k], rest[k], it))
    sv, rv = sum(v for k, v in stage.items() if k in VALU), sum(v for k, v in rest.items() if k in VALU)
    out.append("%-52s %8d %8d %12d" % ("all instructions", sum(stage.values()), sum(rest.values()), sum(rest.values()) + (N - 1) * sum(stage.values())))
    out.append("%-52s %8d %8d %12d" % ("VALU instructions", sv, rv, rv + (N - 1) * sv))
    out.append("f64 arithmetic / VALU per iteration: %.1f %%   (the rest of the loop body holds branches the row counts of a node decide: an upper bound of what executes)" % (100.0 * tot["f64 arithmetic"] / (rv + (N - 1) * sv)))
    res = "\n".join(out); print(res)
    if len(sys.argv) > 1: open(sys.argv[1], "w").write(res + "\n")


if __name__ == "__main__":
    main()
