"""Build-time screen for the kernels that read LDS through inline asm next to LDS-DMA without carrying their wait in the same asm statement
(gemm_tn_kernel_tr / gemm_nn_kernel_m16p in gemm.hip: their phase schedule puts the DMA issue between a group of reads and its wait).
hipcc does not know such a read's result is pending: nothing stops it from copying or using the destination registers before the
`s_waitcnt lgkmcnt(0)` that follows (it did exactly that in an attention kernel: v_mov of registers still in flight, stale data only on a busy
chip).  This script compiles the file to gfx950 assembly and checks that between every `ds_read_b64_tr_b16 vDST, ...` written by inline asm
and the next `s_waitcnt` with lgkmcnt(0) no instruction names a register of vDST.  Exit code 1 and the offending lines otherwise.

    python scripts/check_asm_lds_reads.py [file.hip ...]        (default: ecg_byte_amd/csrc/gemm.hip)
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form"]


def regs_of(tok):
    """'v[12:15]' -> {12..15}; 'v7' -> {7}"""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def vregs_in(line):
    out = set()
    for tok in re.findall(r"v\[\d+:\d+\]|\bv\d+\b", line):
        out |= regs_of(tok)
    return out


def check(asm_text):
    """Returns [(kernel, line number, text, registers)] of uses of a pending inline-asm read's destination."""
    bad, kernel, pending, in_asm = [], "?", {}, False          # pending: register -> line of the read that will write it
    for n, raw in enumerate(asm_text.splitlines(), 1):
        line = raw.split(";")[0].strip() if not raw.strip().startswith(";;#") else raw.strip()
        if raw.strip().startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if raw.strip().startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(_Z\w+):", raw)
        if m:
            kernel, pending = m.group(1), {}
            continue
        if not line or line.endswith(":") or line.startswith("."):
            continue
        if line.startswith("s_waitcnt") and ("lgkmcnt(0)" in line or line.strip() == "s_waitcnt 0"):
            pending = {}
            continue
        if in_asm and line.startswith("ds_read_b64_tr_b16"):
            ops = [t.strip() for t in line[len("ds_read_b64_tr_b16"):].split(",")]
            for r in regs_of(ops[0]):
                pending[r] = n
            continue
        if pending:
            hit = vregs_in(line) & set(pending)
            if hit:
                bad.append((kernel, n, line, sorted(hit)))
    return bad


def main(files=None):
    files = files or [os.path.join(ROOT, "ecg_byte_amd", "csrc", "gemm.hip")]
    rc = 0
    for f in files:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "k.s")
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *FLAGS, f"-I{ROOT}/include", f"-I{ROOT}/ecg_byte_amd/csrc", "-S", "--cuda-device-only",
                            "-o", out, f], check=True, stderr=subprocess.DEVNULL)
            text = open(out).read()
        bad = check(text)
        n_reads = text.count("ds_read_b64_tr_b16")
        print(f"{os.path.relpath(f, ROOT)}: {n_reads} transposing reads, {len(bad)} uses of a register still in flight")
        for kernel, n, line, regs in bad[:20]:
            print(f"   {kernel[:60]} line {n}: {line}   (v{regs})")
        rc |= bool(bad)
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
