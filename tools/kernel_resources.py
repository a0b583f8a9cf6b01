"""Static resource audit of every device kernel in libenvidr_amd.so (no GPU needed): the gfx950 code objects are cut out of the library's
.hip_fatbin section and their kernel descriptors' metadata (llvm-readelf --notes: amdhsa.kernels) listed -- registers (arch VGPRs, AGPRs,
SGPRs), LDS, scratch, spills, workgroup size -- with the kernels that use scratch or spill named.

    python tools/kernel_resources.py [library.so] > profiles/<round>/kernel_resources.txt
    python tools/kernel_resources.py --loops      # additionally: every translation unit compiled to assembly (product flags) and, for each
                                                  # kernel that spills, where the scratch instructions sit -- inside a loop or outside
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
LLVM = Path("/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib: Path):
    """(triple, bytes) of every bundle entry in the library (each translation unit contributes one bundle: a host stub + one gfx950 object)"""
    data = lib.read_bytes()
    out, at = [], data.find(MAGIC)
    while at >= 0:
        n = int.from_bytes(data[at + 24:at + 32], "little")
        p = at + 32
        for _ in range(n):
            off, size, tlen = (int.from_bytes(data[p + 8 * i:p + 8 * i + 8], "little") for i in range(3))
            triple = data[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx" in triple and size:
                out.append((triple, data[at + off:at + off + size]))
        at = data.find(MAGIC, at + 1)
    return out


def kernels_of(blob: bytes):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(blob)
        f.flush()
        notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", f.name], capture_output=True, text=True).stdout
    ks, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s+-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip().strip("'")
        if k == "agpr_count" or (k == "args" and cur is None):
            pass
        if re.match(r"\s+- \.", line) and k in ("agpr_count", "args"):          # first key of a kernel entry (keys are sorted)
            cur = {}
            ks.append(cur)
        if cur is not None and k in ("agpr_count", "vgpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "vgpr_spill_count",
                                     "sgpr_spill_count", "max_flat_workgroup_size", "name", "uses_dynamic_stack"):
            cur[k] = v
    return [k for k in ks if "name" in k]


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(n.replace("DF16_", "Dh") for n in names), capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"\(anonymous namespace\)::", "", x) for x in r]


def loop_report():
    """scratch_load / scratch_store instructions of every kernel, split into those inside a loop (a basic block spanned by a backward
    branch) and outside, with the matrix instructions of the same loops beside them"""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, str(ROOT))
    from envidr_amd import build as B
    out = Path(tempfile.mkdtemp(prefix="kres_"))

    def asm(src):
        dst = out / (src.stem + ".s")
        subprocess.run([B.hipcc(), *[f for f in B.HIPCC_FLAGS if f != "-fPIC"], "-S", "--cuda-device-only", "-o", str(dst), str(src)], capture_output=True, text=True)
        return dst
    with ThreadPoolExecutor(8) as ex:
        files = list(ex.map(asm, B.sources()))
    rows = []
    for f in files:
        if not f.exists():
            continue
        text = f.read_text()
        for m in re.finditer(r"^(_Z\S+|[A-Za-z_]\w*): +; @", text, re.M):
            name = m.group(1)
            end = text.find(".Lfunc_end", m.end())
            lines = text[m.end():end].splitlines()
            label_at, ops = {}, []                       # label -> instruction index; (index, kind)
            branches = []
            n = 0
            for l in lines:
                lm = re.match(r"^(\.LBB\d+_\d+):", l)
                if lm:
                    label_at[lm.group(1)] = n
                    continue
                s = l.strip()
                if not s or s[0] in ";.":
                    continue
                if s.startswith("scratch_"):
                    ops.append((n, "scratch"))
                elif "v_mfma" in s:
                    ops.append((n, "mfma"))
                bm = re.match(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", s)
                if bm:
                    branches.append((n, bm.group(1)))
                n += 1
            loops = [(label_at[t], at) for at, t in branches if t in label_at and label_at[t] <= at]
            inside = lambda i: any(a <= i <= b for a, b in loops)
            sc_in = sum(1 for i, k in ops if k == "scratch" and inside(i))
            sc_out = sum(1 for i, k in ops if k == "scratch" and not inside(i))
            if sc_in or sc_out:
                hot = [(a, b) for a, b in loops if any(k == "scratch" and a <= i <= b for i, k in ops)]
                mf = sum(1 for i, k in ops if k == "mfma" and any(a <= i <= b for a, b in hot))
                rows.append((name, sc_in, sc_out, mf, n))
    names = demangle([r[0] for r in rows])
    print("kernels with scratch instructions: where they sit (a loop = the span of a backward branch)")
    print(f"{'in loops':>9} {'outside':>8} {'MFMAs in those loops':>21} {'instructions':>13}  kernel")
    for (raw, a, b, mf, n), pretty in sorted(zip(rows, names), key=lambda r: -r[0][1]):
        print(f"{a:9d} {b:8d} {mf:21d} {n:13d}  {re.sub(r'^void ', '', pretty)[:150]}")


def main():
    if "--loops" in sys.argv:
        loop_report()
        return
    lib = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "envidr_amd" / "libenvidr_amd.so"
    rows = []
    for triple, blob in code_objects(lib):
        rows += kernels_of(blob)
    names = demangle([r["name"] for r in rows])
    for r, n in zip(rows, names):
        r["pretty"] = re.sub(r"^void ", "", n)
    I = lambda r, k: int(r.get(k, 0) or 0)
    print(f"{lib.name}: {len(rows)} device kernels for gfx950")
    scratch = [r for r in rows if I(r, "private_segment_fixed_size") > 0 or r.get("uses_dynamic_stack") in ("true", "1")]
    spills = [r for r in rows if I(r, "vgpr_spill_count") > 0 or I(r, "sgpr_spill_count") > 0]
    agpr = [r for r in rows if I(r, "agpr_count") > 0]
    print(f"kernels with scratch memory: {len(scratch)}; with register spills: {len(spills)}; using AGPRs: {len(agpr)}")
    print(f"largest: {max(I(r, 'vgpr_count') for r in rows)} vector registers (the unified file: arch VGPRs + AGPRs), of them {max(I(r, 'agpr_count') for r in rows)} AGPRs, {max(I(r, 'sgpr_count') for r in rows)} SGPRs, "
          f"{max(I(r, 'group_segment_fixed_size') for r in rows)} B of LDS")
    for title, sel in (("scratch", scratch), ("spills", spills)):
        for r in sel:
            print(f"  {title}: {r['pretty'][:150]}  scratch {I(r, 'private_segment_fixed_size')} B, spilled VGPRs {I(r, 'vgpr_spill_count')}, SGPRs {I(r, 'sgpr_spill_count')}")
    print()
    print(f"{'VREG':>5} {'AGPR':>5} {'SGPR':>5} {'LDS B':>7} {'scratch':>7} {'wg':>5}  kernel")
    for r in sorted(rows, key=lambda r: (-I(r, "vgpr_count"), r["pretty"])):
        print(f"{I(r, 'vgpr_count'):5d} {I(r, 'agpr_count'):5d} {I(r, 'sgpr_count'):5d} {I(r, 'group_segment_fixed_size'):7d} {I(r, 'private_segment_fixed_size'):7d} "
              f"{I(r, 'max_flat_workgroup_size'):5d}  {r['pretty'][:170]}")


if __name__ == "__main__":
    main()
