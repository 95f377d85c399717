"""Static view of the on-chip interior point kernel's code: compiles csrc/miqp_gpu.hip to gfx950 assembly with phase
markers (-DMIQP_ISA_MARKS) and counts, per phase of ipm_onchip_kernel<2, 10>, the instructions by class - in particular
scratch (spill) traffic and full memory waits inside the iteration loop.  python tools/isa_report.py [out.txt]"""
import collections
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "..", "planner_miqp_amd", "csrc", "miqp_gpu.hip")


def main():
    tmp = "/tmp/isa_report"
    os.makedirs(tmp, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-fno-math-errno", "-freciprocal-math", "-fno-signed-zeros", "-fno-trapping-math", "-std=c++17"]
    if not os.environ.get("ISA_REUSE"): subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-DMIQP_ISA_MARKS", "--cuda-device-only", "-S", "-o", tmp + "/k.s", SRC])
    txt = open(tmp + "/k.s").read()
    sym = "_ZN4miqp17ipm_onchip_kernelILi2ELi10ELi0ELi128EEEvNS_6DevBufE"
    a = txt.index("\n" + sym + ":"); b = txt.index(".Lfunc_end", a)
    body = txt[a:b].split("\n")[2:]
    phase = "prologue"; order = [phase]
    cnt = collections.defaultdict(collections.Counter)
    for ln in body:
        t = ln.strip()
        mk = re.match(r"; OCMARK (\w+)", t)
        if mk:
            phase = mk.group(1)
            if phase not in order: order.append(phase)
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        c = cnt[phase]
        c["instr"] += 1
        if op.startswith("scratch_load"): c["scratch_ld"] += 1
        elif op.startswith("scratch_store"): c["scratch_st"] += 1
        elif op.startswith("v_mfma"): c["mfma"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith("global_") or op.startswith("buffer_"): c["global"] += 1
        elif op == "s_waitcnt":
            c["waitcnt"] += 1
            if "vmcnt(0)" in t: c["wait_vm0"] += 1
        elif op.startswith("v_readlane") or op.startswith("v_readfirstlane"): c["readlane"] += 1
        elif "dpp" in t or op.startswith("v_permlane"): c["dpp"] += 1
    names = {"prologue": "prologue", "tp_d0": "decode", "tp_d1": "init/obj", "tp_r0": "row pass 1", "tp_r1": "sweep head", "tp_s0": "phi(next stage)",
             "tp_s1": "T/S chains, part", "tp_s2": "LDL", "tp_s3": "K solve", "tp_s4": "rank update, gains", "tp_s5": "stage tail", "tp_f0": "forward sweep",
             "tp_f1": "step length", "tp_f2": "update", "tp_f3": "loop tail / epilogue"}
    cols = ["instr", "mfma", "lds", "global", "scratch_ld", "scratch_st", "readlane", "dpp", "waitcnt", "wait_vm0"]
    out = ["ipm_onchip_kernel<2,10>: static instruction counts per phase (marker = start of the phase)",
           "%-22s" % "phase" + "".join("%11s" % c for c in cols)]
    for ph in order:
        out.append("%-22s" % names.get(ph, ph) + "".join("%11d" % cnt[ph][c] for c in cols))
    out.append("%-22s" % "total" + "".join("%11d" % sum(cnt[ph][c] for ph in order) for c in cols))
    meta = re.search(r"\.amdhsa_kernel _ZN4miqp17ipm_onchip_kernelILi2ELi10ELi0ELi128EEEvNS_6DevBufE(.*?)\.end_amdhsa_kernel", txt, re.S).group(1)
    for key in ("next_free_vgpr", "next_free_sgpr", "accum_offset", "private_segment_fixed_size", "group_segment_fixed_size"):
        mm = re.search(r"\.amdhsa_%s (\S+)" % key, meta)
        if mm: out.append("%s = %s" % (key, mm.group(1)))
    res = "\n".join(out)
    print(res)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(res + "\n")


if __name__ == "__main__":
    main()
