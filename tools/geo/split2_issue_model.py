"""Static issue model of the fused-pair split-precision kernel (k_env_split2<5, 8>), no GPU needed: the kernel and its timing-only ablation
variants (tools/geo/build_variants.py s2_*) are compiled to gfx950 assembly, the instructions of one round of a wave (the outermost loop)
are counted by class, and priced with what the probes measured on the chip:

  * v_mfma_f32_32x32x16_f16 back to back: 32.2 ticks each (profiles/r05o/mfma_f16_fill_probe.txt, first row);
  * a vector-ALU instruction between them: the first per MFMA is free, every further one costs 2.03 ticks (same file: 6 per MFMA -> 44.4);
  * an LDS instruction: 17.9 ticks of issue, 52 % of it hidden under MFMAs of the same wave (profiles/r05o/coissue_probe.txt, ds row);
  * the second wave of a SIMD hides nothing of the first's (coissue_probe.txt, "same SIMD" columns: A + B beside each other = A + B).

busy = MFMA ticks / (MFMA + exposed vector + exposed LDS ticks): the share of its cycles the matrix pipe can be busy when nothing ever waits
(no barrier, no memory latency, no dependency stall).  The measured kernel can only be below it.

    python tools/geo/split2_issue_model.py > profiles/<round>/split2_issue_model.txt
"""
import collections
import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools" / "geo"))
from envidr_amd import build as B  # noqa: E402
import build_variants as BV  # noqa: E402

T_MFMA, T_VALU, T_LDS, LDS_EXPOSED = 32.2, 2.03, 17.9, 0.48
# shading stage ms of the variants on one box (profiles/r06a/split2_variants*.txt); the heads kernel's 1.5 ms are not this kernel's
MEASURED_MS = {"s2_base": 12.5, "s2_nodma": 10.9, "s2_noprologue": 11.9, "s2_nobarrier": 12.1, "s2_floor": 10.2}
HEADS_MS = 1.5


def classify(t):
    op = t.split()[0]
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    return "salu" if op.startswith("s_") else "other"


def round_mix(asm: str, skipped_prologue=False, kernel=r"k_env_split2ILi5ELi8E"):
    m = re.search(r"^(_Z\S*" + kernel + r"\S*): +; @", asm, re.M)
    body = asm[m.end():asm.find(".Lfunc_end", m.end())]
    label_at, instrs = {}, []
    for line in body.splitlines():
        lm = re.match(r"^(\.LBB\d+_\d+):", line)
        if lm:
            label_at[lm.group(1)] = len(instrs)
            continue
        t = line.strip()
        if t and t[0] not in ";.":
            instrs.append(t.split(";")[0].strip())
    loops = []
    for i, t in enumerate(instrs):
        bm = re.match(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", t)
        if bm and label_at.get(bm.group(1), i + 1) <= i:
            loops.append((label_at[bm.group(1)], i))
    a, b = max(loops, key=lambda ab: ab[1] - ab[0])
    mix = collections.Counter(classify(t) for t in instrs[a:b + 1])
    if skipped_prologue:
        # s2_noprologue / s2_floor run the IDE + layer-1 operand section only in a workgroup's first round: it is the forward branch inside the
        # loop that jumps over the fp64 instructions (the smallest such span); its instructions are not part of a steady-state round
        spans = []
        for i in range(a, b + 1):
            bm = re.match(r"s_cbranch\S*\s+(\.LBB\d+_\d+)", instrs[i])
            if bm and label_at.get(bm.group(1), 0) > i + 50 and any("_f64" in x for x in instrs[i:label_at[bm.group(1)]]):
                spans.append((label_at[bm.group(1)] - i, i, label_at[bm.group(1)]))
        _, i, j = min(spans)
        mix = mix - collections.Counter(classify(t) for t in instrs[i:j])
    return mix, b - a + 1


def main():
    work = Path(tempfile.mkdtemp(prefix="s2model_"))
    (work / "include").symlink_to(ROOT / "include")
    print(__doc__.split("\n\n")[0].replace("\n", " "))
    print(f"prices: MFMA {T_MFMA}, vector instruction beyond one per MFMA {T_VALU}, LDS instruction {T_LDS} x {LDS_EXPOSED} exposed (ticks)\n")
    print(f"{'variant':<14} {'MFMA':>5} {'vector':>7} {'LDS':>5} {'VMEM':>5} {'scalar':>7} {'waits':>6} {'barriers':>8} | {'model busy':>10} | {'measured kernel ms':>18} {'busy at 6.6 ms of MFMA':>23}")
    for name in MEASURED_MS:
        unit, patches = BV.VARIANTS[name]
        src = work / name / "csrc"
        shutil.copytree(B.CSRC, src, ignore=shutil.ignore_patterns("build", "*.o"))
        for fname, old, new in patches:
            text = (src / fname).read_text()
            assert text.count(old) == 1, (name, fname, old)
            (src / fname).write_text(text.replace(old, new))
        out = work / f"{name}.s"
        r = subprocess.run([B.hipcc(), *[f for f in B.HIPCC_FLAGS if f != "-fPIC"], "-S", "--cuda-device-only", "-o", str(out), str(src / f"{unit}.hip")],
                           capture_output=True, text=True)
        if r.returncode:
            print(name, "did not compile:", r.stderr[-500:])
            continue
        c, n = round_mix(out.read_text(), skipped_prologue=name in ("s2_noprologue", "s2_floor"))
        t_m = c["mfma"] * T_MFMA
        t_v = max(c["valu"] - c["mfma"], 0) * T_VALU
        t_l = c["lds"] * T_LDS * LDS_EXPOSED
        busy = t_m / (t_m + t_v + t_l)
        kernel_ms = MEASURED_MS[name] - HEADS_MS
        print(f"{name:<14} {c['mfma']:5d} {c['valu']:7d} {c['lds']:5d} {c['vmem']:5d} {c['salu']:7d} {c['wait']:6d} {c['barrier']:8d} | {busy:10.3f} | {kernel_ms:18.1f} {6.6 / kernel_ms:23.2f}")
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
