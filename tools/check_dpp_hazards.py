"""Static check of the gfx950 ISA of csrc/solve.hip: no `v_fmac_f32_dpp` / `v_fmac_f64_dpp` of k_spmm_sym_bcast reads (as its DPP
operand, src0) a VGPR that a VALU instruction wrote fewer than two wait states earlier, none follows a write of EXEC (v_cmpx, or any
instruction whose destination is exec) by fewer than five, and the kernels neither spill nor use AGPR copies.  The FMAs are inline asm (csrc/spmm_sym_bcast.h), which hipcc's hazard recogniser does not look into.
Usage: python tools/check_dpp_hazards.py [path/to/solve.s]   (without an argument: compiles solve.hip to ISA first, ~25 s)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(txt):
    """{kernel: (dpp_fmas, hazards, spills)} for every k_spmm_sym_bcast instantiation in the ISA text."""
    res = {}
    for name in re.findall(r"^(_Z\d+k_spmm_sym_bcast\w+):", txt, re.M):
        body = txt[txt.index(name + ":"):]
        body = body[:body.index("s_endpgm")]
        hist, bad, n, spills, exec_age = [], 0, 0, 0, 99
        for line in body.split("\n"):
            s = line.strip()
            if not s or s[0] in ";." or s.endswith(":"):
                continue
            op = s.split()[0]
            args = [a.strip() for a in s[len(op):].split(",")]
            if "accvgpr" in op or op.startswith("scratch_"):
                spills += 1
            if op.startswith("v_fmac") and "dpp" in op:
                n += 1
                src0 = _regs(args[1].split()[0])
                bad += sum(1 for ws, w in hist if ws < 2 and (w & src0))
                bad += exec_age < 5
            adv = int(args[0]) + 1 if op == "s_nop" else 1
            hist = [(ws + adv, w) for ws, w in hist if ws + adv < 3]
            exec_age += adv
            if op.startswith("v_cmpx") or (args and args[0].split() and args[0].split()[0].startswith("exec")):
                exec_age = 0
            if op.startswith("v_") and not op.startswith("v_cmp"):
                hist.append((0, _regs(args[0].split()[0])))
        res[name] = (n, bad, spills)
    return res


def main():
    if len(sys.argv) > 1:
        txt = open(sys.argv[1]).read()
    else:
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "solve.s")
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--offload-device-only", "-S", "-o", out,
                            os.path.join(ROOT, "online_gp_amd", "csrc", "solve.hip")], check=True, stderr=subprocess.DEVNULL)
            txt = open(out).read()
    res = scan(txt)
    ok = bool(res)
    for k, (n, bad, spills) in sorted(res.items()):
        print(f"{k}: {n} DPP FMAs, {bad} hazards, {spills} spill / AGPR instructions")
        ok = ok and n > 0 and bad == 0 and spills == 0
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
