"""A/B (round-5 review, item 2 i): the increment select trees of the hot / cold game kernel as e32 `v_cndmask` on VCC — one `v_cmp` per
tree node shared by the four dwords of an increment, emitted as inline-asm groups of four selects the compiler may schedule other work
between — against the shipped form (`(s & H) ? hi : lo`, which the compiler emits as `v_cmp_e64` into an SGPR pair + `v_cndmask_e64`).

The shipped headers stay free of experiments: `build` copies `farkle_ii_amd/csrc` + `include` to exp/vcc_trees/, rewrites `begin_turn`'s
increment pick there, and compiles exp/vcc_trees/libfarkle_hip.so (hipcc cross-compiles without a GPU); `run` (on the GPU box) plays
bench config 6 and the parity subset with each library (`FARKLE_HIP_LIB`) and writes the comparison.

usage: python tools/ab_vcc_select_trees.py build | run OUT.json
"""
import json, os, shutil, subprocess, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
WORK = ROOT / "exp" / "vcc_trees"
LIB = WORK / "libfarkle_hip.so"

HELPER = r'''
// (A/B variant, tools/ab_vcc_select_trees.py) four dwords of entry s of an N-entry table of register quadruples: per tree node one v_cmp
// into VCC and four e32 selects on it, one inline-asm group the compiler may move other instructions around
struct HcQuad { uint32_t x, y, z, w; };
template <int N, int LO, int CNT, typename F>
__device__ __forceinline__ HcQuad hc_pick4_r(uint32_t s, F get) {
    if constexpr (CNT == 1) {
        return get(LO < N ? LO : N - 1);
    } else {
        constexpr int H = CNT / 2;
        if constexpr (LO + H >= N) {
            return hc_pick4_r<N, LO, H>(s, get);
        } else {
            const HcQuad lo = hc_pick4_r<N, LO, H>(s, get), hi = hc_pick4_r<N, LO + H, H>(s, get);
            const uint32_t m = s & (uint32_t)H;
            HcQuad r;
            asm("v_cmp_ne_u32_e32 vcc, 0, %8\n\t"
                "v_cndmask_b32_e32 %0, %4, %9, vcc\n\t"
                "v_cndmask_b32_e32 %1, %5, %10, vcc\n\t"
                "v_cndmask_b32_e32 %2, %6, %11, vcc\n\t"
                "v_cndmask_b32_e32 %3, %7, %12, vcc"
                : "=&v"(r.x), "=&v"(r.y), "=&v"(r.z), "=&v"(r.w)
                : "v"(lo.x), "v"(lo.y), "v"(lo.z), "v"(lo.w), "v"(m), "v"(hi.x), "v"(hi.y), "v"(hi.z), "v"(hi.w)
                : "vcc");
            return r;
        }
    }
}
'''

OLD_PICK = '''            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = hc_pick<NI>(s, [&](int t) __attribute__((always_inline)) { return inc_r[t][j]; });
            inc = make_uint4(w[0], w[1], w[2], w[3]);
'''
NEW_PICK = '''            const HcQuad q = hc_pick4_r<NI, 0, 16>(s, [&](int t) __attribute__((always_inline)) {
                return HcQuad{inc_r[t][0], inc_r[t][1], inc_r[t][2], inc_r[t][3]};
            });
            inc = make_uint4(q.x, q.y, q.z, q.w);
'''


def build() -> None:
    if WORK.exists():
        shutil.rmtree(WORK)
    (WORK / "farkle_ii_amd").mkdir(parents=True)
    shutil.copytree(ROOT / "farkle_ii_amd" / "csrc", WORK / "farkle_ii_amd" / "csrc")
    shutil.copytree(ROOT / "include", WORK / "include")
    hc = WORK / "farkle_ii_amd" / "csrc" / "fk_play_hc.h"
    text = hc.read_text()
    assert text.count(OLD_PICK) == 1
    anchor = "// KI: seats whose PCG increments"
    assert text.count(anchor) == 1
    hc.write_text(text.replace(OLD_PICK, NEW_PICK).replace(anchor, HELPER + "\n" + anchor))
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", str(LIB),
           str(WORK / "farkle_ii_amd" / "csrc" / "farkle_hip.hip")]
    subprocess.run(cmd, check=True)
    # what the compiler made of it: e32 selects on vcc in the game kernels
    dis = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o", f"--input={LIB}"], capture_output=True, text=True)
    print("built", LIB, dis.stdout.strip().splitlines()[:2])


def run(out_path: str) -> None:
    results = {"what": __doc__.split("\n\n")[0], "variants": {}}
    for name, lib in (("shipped", None), ("vcc_e32_trees", LIB), ("shipped_again", None)):
        env = dict(os.environ)
        if lib is not None:
            env["FARKLE_HIP_LIB"] = str(lib)
        tests = subprocess.run([sys.executable, "-m", "pytest", "tests/test_hot_cold_gpu.py", "-x", "-q", "-m", "gpu"], cwd=ROOT, env=env,
                               capture_output=True, text=True) if name != "shipped_again" else None  # every hot / cold instance family against the oracle
        bench = subprocess.run([sys.executable, "bench.py", "--config", "6", "--steps", "2", "--warmup", "1"], cwd=ROOT, env=env, capture_output=True, text=True)
        line = json.loads(bench.stdout.strip().splitlines()[-1])
        per_k = {str(e["k"]): {"kernel_ms": e.get("kernel_ms"), "frac": e.get("frac")} for e in line.get("roofline", {}).get("per_k", [])} if isinstance(line.get("roofline", {}).get("per_k"), list) else line.get("roofline")
        results["variants"][name] = {"library": str(lib) if lib else "farkle_ii_amd/libfarkle_hip.so", "parity_tests": None if tests is None else tests.stdout.strip().splitlines()[-1],
                                     "value_games_per_s": line["value"], "ms_per_step": line["ms_per_step"], "roofline": per_k}
        print(name, results["variants"][name]["parity_tests"], line["value"], flush=True)
    Path(out_path).write_text(json.dumps(results, indent=1))


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run(sys.argv[2])
