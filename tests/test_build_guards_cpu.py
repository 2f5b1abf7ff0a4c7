"""CPU guards on the hand-written idioms the bit-exact GPU tests depend on, run on every build instead of by hand:

* the scalar-load idiom of csrc/emission.h (an `s_load_dwordx8` whose `s_waitcnt` sits in a separate asm statement) —
  `sapr_amd.build` scans the assembly of every translation unit that uses it and records the result;
* the four-instruction exactly-rounded division of the emission kernels against IEEE division on the CPU
  (scripts/verify/fastdiv_check.c, a reduced operand count here);
* the E-step's exp / reciprocal / log(1 + e) chain (csrc/lse_unit.h, plain C) against expl / log1pl
  (scripts/verify/lse_unit_check.c);
* the packing of the banded mel filterbank into the 16 blocks of `v_mfma_f32_4x4x1_16b_f32` (csrc/mfcc_wave_pack.h,
  plain C++): every filter weight lands exactly once, block starts make the B-operand reads conflict-free."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_scalar_load_scan_ran_on_every_user_of_the_idiom_and_found_nothing():
    from sapr_amd import build
    build.build(verbose=False)
    with open(build.SCAN_RECORD) as fh:
        rec = json.load(fh)
    users = [s for s in build.SOURCES if build.uses_sload_idiom(s)]
    assert users and sorted(rec) == sorted(users)
    assert all(r["violations"] == 0 for r in rec.values())
    # the exact Viterbi kernels, the bounding pass and the E-step really contain hand-written scalar loads
    for src in users:
        if src.startswith(("viterbi_exact", "viterbi_bound", "estep")):
            assert rec[src]["loads"] > 0, src
    assert sum(r["loads"] for r in rec.values()) > 1000


def test_scanner_flags_a_read_between_load_and_wait(tmp_path):
    from sapr_amd.asm_scan import check
    good = tmp_path / "good.s"
    good.write_text(";;#ASMSTART\n\ts_load_dwordx8 s[8:15], s[2:3], 0x0\n;;#ASMEND\n\tv_add_f64 v[0:1], v[2:3], v[4:5]\n"
                    ";;#ASMSTART\n\ts_waitcnt lgkmcnt(0)\n;;#ASMEND\n\tv_fma_f64 v[0:1], s[8:9], v[2:3], v[0:1]\n\ts_endpgm\n")
    bad = tmp_path / "bad.s"
    bad.write_text(";;#ASMSTART\n\ts_load_dwordx8 s[8:15], s[2:3], 0x0\n;;#ASMEND\n\ts_mov_b64 s[20:21], s[10:11]\n"
                   ";;#ASMSTART\n\ts_waitcnt lgkmcnt(0)\n;;#ASMEND\n\ts_endpgm\n")
    assert check(str(good), verbose=False) == (1, 0)
    assert check(str(bad), verbose=False) == (1, 1)


def test_scanner_follows_control_flow_and_flags_writes(tmp_path):
    """Between a hand-written scalar load and its wait the scanner also refuses a compiler-inserted WRITE of the
    destination registers (the late load would overwrite it), and it walks every path of the control-flow graph: an
    if-block (in line or moved out of line) is followed on both sides; a loop back to the load, or a path that skips the
    wait and then uses the registers, is reported."""
    from sapr_amd.asm_scan import check
    LOAD = ";;#ASMSTART\n\ts_load_dwordx8 s[8:15], s[2:3], 0x0\n;;#ASMEND\n"
    WAIT = ";;#ASMSTART\n\ts_waitcnt lgkmcnt(0)\n;;#ASMEND\n"
    END = "\tv_fma_f64 v[0:1], s[8:9], v[2:3], v[0:1]\n\ts_endpgm\n"
    cases = {
        "write": (LOAD + "\ts_mov_b32 s9, 0\n" + WAIT + END, 1),
        "if_block": (LOAD + "\ts_and_saveexec_b64 s[20:21], vcc\n\ts_cbranch_execz .LBB0_2\n\tglobal_store_dwordx2 v[0:1], "
                     "v[2:3], off\n.LBB0_2:\n\ts_or_b64 exec, exec, s[20:21]\n" + WAIT + END, 0),
        "if_block_reads": (LOAD + "\ts_cbranch_execz .LBB0_2\n\tv_mov_b32_e32 v9, s12\n.LBB0_2:\n" + WAIT + END, 1),
        "out_of_line": (LOAD + "\ts_cbranch_execnz .LBB0_9\n.LBB0_2:\n" + WAIT + END +
                        ".LBB0_9:\n\tv_add_f64 v[4:5], v[4:5], v[6:7]\n\ts_branch .LBB0_2\n", 0),
        "out_of_line_reads": (LOAD + "\ts_cbranch_execnz .LBB0_9\n.LBB0_2:\n" + WAIT + END +
                              ".LBB0_9:\n\tv_writelane_b32 v7, s14, 3\n\ts_branch .LBB0_2\n", 1),
        "loop": (".LBB0_1:\n" + LOAD + "\ts_cbranch_scc1 .LBB0_1\n" + WAIT + END, 1),
        "past_wait": (LOAD + "\ts_cbranch_execz .LBB0_3\n" + WAIT + ".LBB0_3:\n" + END, 1),
        "indirect": (LOAD + "\ts_setpc_b64 s[30:31]\n" + WAIT + END, 1),
        # long-branch relaxation = a jump to a known label: followed like s_branch (here into a block that reads s13)
        "far_jump": (LOAD + "\ts_cbranch_execz .LBB0_2\n\ts_getpc_b64 s[18:19]\n.Lpost_getpc1:\n\ts_add_u32 s18, s18, "
                     "(.LBB0_9-.Lpost_getpc1)&4294967295\n\ts_addc_u32 s19, s19, (.LBB0_9-.Lpost_getpc1)>>32\n"
                     "\ts_setpc_b64 s[18:19]\n.LBB0_2:\n" + WAIT + END + ".LBB0_9:\n\ts_mov_b32 s40, s13\n\ts_branch .LBB0_2\n", 1),
    }
    for name, (text, want) in cases.items():
        f = tmp_path / f"{name}.s"
        f.write_text(text)
        assert check(str(f), verbose=False) == (1, want), name


def test_fast_division_chain_equals_ieee_division(tmp_path):
    exe = tmp_path / "fastdiv_check"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-o", str(exe),
                           os.path.join(ROOT, "scripts", "verify", "fastdiv_check.c"), "-lm"])
    out = subprocess.run([str(exe), "40000000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mismatches 0 " in out.stdout, out.stdout


def test_log_sum_exp_terms_of_the_lattice_recursions_stay_within_ulps(tmp_path):
    """csrc/lse_unit.h: e = exp(-d), 1 / (1 + e) and log(1 + e) from one short chain (a Taylor exponential, one
    reciprocal by Newton steps from a float32-accurate estimate, the atanh series) against expl / log1pl over d in
    [0, 760], the reduction boundaries, e -> 1, the clamp, infinity and NaN."""
    exe = tmp_path / "lse_unit_check"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe),
                           os.path.join(ROOT, "scripts", "verify", "lse_unit_check.c"), "-lm"])
    out = subprocess.run([str(exe), "2000000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "edges ok" in out.stdout, out.stdout


PACK_MAIN = r'''
#include "mfcc_wave_pack.h"
#include <cstdio>
int main(int argc, char **argv) {
  int n_mels, nb;
  if (std::scanf("%d %d", &n_mels, &nb) != 2) return 2;
  std::vector<float> w(static_cast<size_t>(n_mels) * nb);
  for (auto &v : w) if (std::scanf("%f", &v) != 1) return 2;
  WavePack wp = wave_pack(w, n_mels, nb);
  std::printf("%d %d\n", wp.s4, wp.conflict_free_passes);
  for (int v : wp.blk) std::printf("%d ", v);
  std::printf("\n");
  for (float v : wp.a) std::printf("%.9g ", v);
  std::printf("\n");
  return 0;
}
'''


@pytest.mark.parametrize("sr,n_mels", [(16000, 40), (8000, 24), (16000, 32), (16000, 36), (16000, 20), (16000, 64)])
def test_filterbank_block_packing(tmp_path, sr, n_mels):
    from oracle import mfcc_oracle as mo
    src = tmp_path / "pack_main.cpp"
    src.write_text(PACK_MAIN)
    exe = tmp_path / "pack_main"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "sapr_amd", "csrc"), str(src), "-o",
                           str(exe)])
    M = mo.mel_filterbank(sr, 512, n_mels)
    text = f"{n_mels} 257\n" + " ".join(f"{v:.9g}" for v in M.reshape(-1))
    out = subprocess.run([str(exe)], input=text, capture_output=True, text=True, check=True).stdout.split("\n")
    s4, free = (int(v) for v in out[0].split())
    if n_mels in (20, 64):   # a group's band longer than four blocks' worth of steps, or more parts than blocks:
        assert s4 == 0       # no packing, the plan keeps the workgroup-tile core
        return
    assert 6 <= s4 <= 8, "this filterbank should fit the instantiated step counts"
    blk = np.array(out[1].split(), dtype=np.int64).reshape(16, 4)
    a = np.array(out[2].split(), dtype=np.float32).reshape(s4, 64, 4)
    # rebuild the filterbank from the packed operands: weight of (block b, row i) at step 4 q + c belongs to mel
    # head(b) + i and bin k0_b + 4 q + c; parts of a group follow their head on the next blocks of the same row
    rebuilt = np.zeros_like(M)
    head = np.full(16, -1)
    for b in range(16):
        h = b
        while blk[h, 1] < 0 and h % 4 > 0 and blk[h - 1, 2]:
            h -= 1
        head[b] = blk[h, 1]
    for q in range(s4):
        for lane in range(64):
            b, i = divmod(lane, 4)
            for c in range(4):
                v = a[q, lane, c]
                if v != 0:
                    assert head[b] >= 0
                    rebuilt[head[b] + i, blk[b, 0] + 4 * q + c] += v
    np.testing.assert_array_equal(rebuilt, M)
    assert (blk[:, 0] % 4 == 0).all() and (blk[:, 0] >= 0).all() and (blk[:, 0] + 4 * s4 <= 272).all()
    assert free == 4, "B-operand reads of the four ds_read_b128 passes should be conflict-free for these presets"
    for grp in ([0, 3, 5, 6], [1, 2, 4, 7], [8, 11, 13, 14], [9, 10, 12, 15]):
        assert len({(blk[b, 0] // 4) % 4 for b in grp}) == 4
