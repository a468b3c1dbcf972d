"""Static vector-instruction mix of the game kernels' roll loop, priced with the two issue classes of profiles/r05_valu_issue_rates.txt.
usage: python tools/isa_mix.py [out.json]      (compiles the device code to assembly first: ~30 s)
For every game-kernel instance: the Depth=2 loop with the most v_mad_u64_u32 (the roll step: three PCG64DXSM draws) and everything nested
in it; its VALU instructions by class (full rate ~2.3 cycles per wave-instruction per SIMD, half rate ~4.2), the mean issue cost of the mix
and the fraction of the nominal 2-cycle peak a SIMD that issues this mix back to back would reach (`ceiling_frac` = 2 / mean).  Static:
rarely executed blocks inside the loop (Lemire replay, error exits) are counted like the others.  bench.py attaches the entry of the
instance a configuration runs as roofline.mix while the sha256 of the kernel sources still matches."""
import json, re, subprocess, sys
from collections import Counter
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from bench import kernel_source_sha

FAST = 2.3   # measured 2.2-2.4 at 6-8 waves per SIMD, 2.4-2.5 at 4
SLOW = 4.2   # measured 4.1-4.3
FAST_OPS = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32", "v_lshrrev_b32",
            "v_ashrrev_i32", "v_fma_f32", "v_add_f32", "v_mul_f32", "v_accvgpr", "v_nop")


def is_fast(op: str) -> bool:
    if op.startswith("v_cndmask_b32"):
        return op.endswith("_e32")  # on VCC; the e64 form (mask in an SGPR pair) is half rate
    if op.endswith("_sdwa") or op.endswith("_dpp"):
        return False
    base = op[:-4] if op.endswith(("_e32", "_e64")) else op
    return base in FAST_OPS


def kernels(path):
    name, body = None, []
    for line in open(path):
        if line.startswith("_ZN") and "fk_play" in line and re.match(r"^_ZN\w+:", line):
            name, body = line.split(":")[0], []
        elif name is not None:
            body.append(line.rstrip("\n"))
            if "s_endpgm" in line:
                yield name, body
                name = None


def pretty(sym: str) -> str:
    m = re.search(r"(fk_play(?:_hc)?_kernel)I(.*?)EEvNS", sym)
    args = re.findall(r"L([ijb])(\d+)E", m.group(2))
    return f"{m.group(1)}<{', '.join(v for _, v in args)}>"


def roll_loop(body):
    # blocks: (name, annotation text, instructions); a block starts at a label or at a fall-through marker "; %bb.N:"
    fn = next(re.match(r"\.LBB(\d+)_", l).group(1) for l in body if l.startswith(".LBB"))
    blocks, name, ann, ins = [], None, "", []
    for l in body:
        m = re.match(r"^\.L(BB\d+_\d+):", l) or re.match(r"^; %bb\.(\d+):", l)
        if m:
            if name:
                blocks.append((name, ann, ins))
            name = m.group(1) if m.group(1).startswith("BB") else f"BB{fn}_{m.group(1)}"
            ann, ins = l, []
        elif name and l.lstrip().startswith(";") and not ins:
            ann += " " + l
        elif name and l.startswith("\t") and not l.strip().startswith((";", ".")):
            ins.append(l.split()[0])
    if name:
        blocks.append((name, ann, ins))
    best, best_n = None, -1
    for h, a, _ in blocks:
        if "Loop Header: Depth=2" not in a:
            continue
        members = [b for b in blocks if b[0] == h or re.search(rf"(Header=|Parent Loop ){h}\b", b[1])]
        ops = Counter(o for b in members for o in b[2])
        if ops.get("v_mad_u64_u32", 0) > best_n:
            best, best_n = ops, ops.get("v_mad_u64_u32", 0)
    return best


def main():
    out_path = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "profiles" / "r05_isa_mix.json"
    asm = ROOT / "tools" / "ab" / "farkle_hip_gfx950.s"
    asm.parent.mkdir(exist_ok=True)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", str(ROOT / "include"), "-S", "--cuda-device-only", "-o", str(asm),
                    str(ROOT / "farkle_ii_amd" / "csrc" / "farkle_hip.hip")], check=True, capture_output=True)
    res = {}
    for sym, body in kernels(asm):
        ops = roll_loop(body)
        if not ops:
            continue
        valu = {o: n for o, n in ops.items() if o.startswith("v_") and not o.startswith(("v_readlane", "v_readfirstlane", "v_writelane"))}
        n = sum(valu.values())
        fast = sum(c for o, c in valu.items() if is_fast(o))
        mean = (fast * FAST + (n - fast) * SLOW) / n
        res[pretty(sym)] = {"valu_static": n, "full_rate": fast, "half_rate": n - fast, "mean_issue_cycles": round(mean, 3),
                            "ceiling_frac": round(2.0 / mean, 4), "v_mad_u64_u32": ops.get("v_mad_u64_u32", 0),
                            "v_cndmask": sum(c for o, c in valu.items() if o.startswith("v_cndmask")),
                            "v_mov_b32": sum(c for o, c in valu.items() if o.startswith("v_mov_b32")),
                            "salu_static": sum(c for o, c in ops.items() if o.startswith("s_") and not o.startswith(("s_nop", "s_waitcnt", "s_cbranch", "s_branch"))),
                            "lds": sum(c for o, c in ops.items() if o.startswith("ds_")),
                            "vmem": sum(c for o, c in ops.items() if o.startswith(("global_", "buffer_", "flat_", "scratch_")))}
    doc = {"what": __doc__.split("usage")[0].strip(), "classes": {"full_rate_cycles": FAST, "half_rate_cycles": SLOW, "source": "profiles/r05_valu_issue_rates.txt"},
           "kernel_source_sha256": kernel_source_sha(), "instances": res}
    out_path.write_text(json.dumps(doc, indent=1) + "\n")
    for k in sorted(res):
        r = res[k]
        print(f"{k:60s} VALU {r['valu_static']:4d} full {r['full_rate']:4d} half {r['half_rate']:4d} mean {r['mean_issue_cycles']:.2f} ceiling {r['ceiling_frac']:.3f} "
              f"mad64 {r['v_mad_u64_u32']} cndmask {r['v_cndmask']} mov {r['v_mov_b32']}")


main()
