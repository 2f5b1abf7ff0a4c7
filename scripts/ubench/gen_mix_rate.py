"""Generates scripts/ubench/mix_rate.hip: issue cost of the instruction classes of mfcc_wave_kernel's set body, one
class at a time and as the kernel's own mix, at 1 / 2 / 4 wavefronts per SIMD (dev tool; DESIGN §6 'issue-cost model').

Every test is ONE asm statement: a counted loop of 64 instructions on independent registers, bracketed by s_memtime
(shader-clock cycles of the wavefront itself), so neither the compiler nor the clock governor enters the figure.
    python scripts/ubench/gen_mix_rate.py && hipcc -O2 --offload-arch=gfx950 scripts/ubench/mix_rate.hip -o scripts/ubench/mix_rate
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def body(kind, n=64):
    out = []
    for i in range(n):
        k, j = i % 16, (i + 5) % 16
        if kind == "v_add_f32":
            out.append(f"v_add_f32 v{k}, v{k}, v{16 + j}")
        elif kind == "v_sub_f32":
            out.append(f"v_sub_f32 v{k}, v{16 + j}, v{k}")
        elif kind == "v_mul_f32":
            out.append(f"v_mul_f32 v{k}, v{k}, v{16 + j}")
        elif kind == "v_fma_f32":
            out.append(f"v_fma_f32 v{k}, v{k}, v{16 + j}, v{16 + k}")
        elif kind == "v_fmac_f32":
            out.append(f"v_fmac_f32 v{k}, v{16 + j}, v{16 + k}")
        elif kind == "v_mov_b32_dpp":
            out.append(f"v_mov_b32_dpp v{k}, v{16 + j} row_shr:1 row_mask:0xf bank_mask:0xf")
        elif kind == "v_add_f32_dpp":
            out.append(f"v_add_f32_dpp v{k}, v{16 + j}, v{k} row_shl:1 row_mask:0xf bank_mask:0xf")
        elif kind == "v_cndmask_b32_dpp":
            out.append(f"v_cndmask_b32_dpp v{k}, v{k}, v{16 + j}, vcc row_mirror row_mask:0xf bank_mask:0xf")
        elif kind == "v_cndmask_b32":
            out.append(f"v_cndmask_b32 v{k}, v{k}, v{16 + j}, vcc")
        elif kind == "v_cndmask_b32_dpp_smov":  # as mfcc_wave.h mirror_unless2: vcc re-written by the SALU in front of every pair
            if i % 2 == 0:
                out.append("s_mov_b64 vcc, s[22:23]")
                out.append("s_nop 0")
            out.append(f"v_cndmask_b32_dpp v{k}, v{k}, v{16 + j}, vcc row_mirror row_mask:0xf bank_mask:0xf")
        elif kind == "v_cndmask_b32_vcc_salu":  # vcc written ONCE by the SALU, before the loop
            if i == 0:
                out.append("s_mov_b64 vcc, s[22:23]")
                out.append("s_nop 4")
            out.append(f"v_cndmask_b32 v{k}, v{k}, v{16 + j}, vcc")
        elif kind == "v_cndmask_b32_sgpr":     # VOP3 form, mask in an SGPR pair that the SALU wrote
            out.append(f"v_cndmask_b32_e64 v{k}, v{k}, v{16 + j}, s[22:23]")
        elif kind == "v_cndmask_b32_indep":    # destination differs from both sources (no read-modify-write)
            out.append(f"v_cndmask_b32 v{k}, v{16 + k}, v{16 + j}, vcc")
        elif kind == "v_bfi_b32":
            out.append(f"v_bfi_b32 v{k}, v35, v{16 + j}, v{k}")
        elif kind == "v_and_or_b32":
            out.append(f"v_and_or_b32 v{k}, v{16 + j}, v35, v{k}")
        elif kind == "v_log_f32":
            out.append(f"v_log_f32 v{k}, v{16 + j}")
        elif kind == "v_mov_b32":
            out.append(f"v_mov_b32 v{k}, v{16 + j}")
        elif kind == "v_pk_mul_f32":
            out.append(f"v_pk_mul_f32 v[{2 * (k % 8)}:{2 * (k % 8) + 1}], v[{2 * (k % 8)}:{2 * (k % 8) + 1}], v[{16 + 2 * (j % 8)}:{17 + 2 * (j % 8)}]")
        elif kind == "v_pk_add_f32":
            out.append(f"v_pk_add_f32 v[{2 * (k % 8)}:{2 * (k % 8) + 1}], v[{2 * (k % 8)}:{2 * (k % 8) + 1}], v[{16 + 2 * (j % 8)}:{17 + 2 * (j % 8)}]")
        elif kind == "v_pk_fma_f32":
            out.append(f"v_pk_fma_f32 v[{2 * (k % 8)}:{2 * (k % 8) + 1}], v[{2 * (k % 8)}:{2 * (k % 8) + 1}], v[{16 + 2 * (j % 8)}:{17 + 2 * (j % 8)}], v[{16 + 2 * (k % 8)}:{17 + 2 * (k % 8)}]")
        elif kind == "ds_read_b128":
            out.append(f"ds_read_b128 v[{4 * (k % 4)}:{4 * (k % 4) + 3}], v33 offset:{(i % 16) * 16 * 64 % 16384}")
            if i % 4 == 3:
                out.append("s_waitcnt lgkmcnt(0)")
        elif kind == "ds_write2_b32":
            out.append(f"ds_write2_b32 v34, v{k}, v{16 + j} offset0:{(i % 2) * 128} offset1:{(i % 2) * 128 + 64}")
            if i % 8 == 7:
                out.append("s_waitcnt lgkmcnt(0)")
        elif kind == "ds_read_b64":
            out.append(f"ds_read_b64 v[{2 * (k % 8)}:{2 * (k % 8) + 1}], v32 offset:{(i % 32) * 8 * 64 % 16384}")
            if i % 8 == 7:
                out.append("s_waitcnt lgkmcnt(0)")
        elif kind == "ds_write_b32":
            out.append(f"ds_write_b32 v34, v{k} offset:{(i % 32) * 4 * 64 % 16384}")
            if i % 8 == 7:
                out.append("s_waitcnt lgkmcnt(0)")
        elif kind == "v_fma_f64":
            out.append(f"v_fma_f64 v[{2 * (k % 8)}:{2 * (k % 8) + 1}], v[{2 * (k % 8)}:{2 * (k % 8) + 1}], v[{16 + 2 * (j % 8)}:{17 + 2 * (j % 8)}], v[{16 + 2 * (k % 8)}:{17 + 2 * (k % 8)}]")
        elif kind == "v_add_f64":
            out.append(f"v_add_f64 v[{2 * (k % 8)}:{2 * (k % 8) + 1}], v[{2 * (k % 8)}:{2 * (k % 8) + 1}], v[{16 + 2 * (j % 8)}:{17 + 2 * (j % 8)}]")
        elif kind == "v_mul_f64":
            out.append(f"v_mul_f64 v[{2 * (k % 8)}:{2 * (k % 8) + 1}], v[{2 * (k % 8)}:{2 * (k % 8) + 1}], v[{16 + 2 * (j % 8)}:{17 + 2 * (j % 8)}]")
        elif kind == "v_fma_f64_sgpr":   # one operand from the scalar file (the coefficient chains of lse_unit.h)
            out.append(f"v_fma_f64 v[{2 * (k % 8)}:{2 * (k % 8) + 1}], v[{2 * (k % 8)}:{2 * (k % 8) + 1}], v[{16 + 2 * (j % 8)}:{17 + 2 * (j % 8)}], s[22:23]")
        elif kind == "v_fmac_f64_dpp":   # src0 from lane i % 13 of each row of 16 lanes (custom.hip emission rows)
            out.append(f"v_fmac_f64_dpp v[{2 * (k % 8)}:{2 * (k % 8) + 1}], v[{16 + 2 * (j % 8)}:{17 + 2 * (j % 8)}], v[{16 + 2 * (k % 8)}:{17 + 2 * (k % 8)}] row_newbcast:{i % 13} row_mask:0xf bank_mask:0xf")
        elif kind == "v_fmac_f64":
            out.append(f"v_fmac_f64 v[{2 * (k % 8)}:{2 * (k % 8) + 1}], v[{16 + 2 * (j % 8)}:{17 + 2 * (j % 8)}], v[{16 + 2 * (k % 8)}:{17 + 2 * (k % 8)}]")
        elif kind == "ds_read2_b64":
            out.append(f"ds_read2_b64 v[{4 * (k % 4)}:{4 * (k % 4) + 3}], v32 offset0:{(i % 16) * 2} offset1:{(i % 16) * 2 + 1}")
            if i % 4 == 3:
                out.append("s_waitcnt lgkmcnt(0)")
        elif kind == "ds_read_b64_bcast":   # every lane reads the same address (custom_emission_exact_kernel's staged frames)
            out.append(f"ds_read_b64 v[{2 * (k % 8)}:{2 * (k % 8) + 1}], v36 offset:{(i % 32) * 8}")
            if i % 8 == 7:
                out.append("s_waitcnt lgkmcnt(0)")
        elif kind == "v_mfma_f32_4x4x1":
            a = 4 * (i % 8)
            out.append(f"v_mfma_f32_4x4x1_16b_f32 a[{a}:{a + 3}], v{k}, v{16 + j}, a[{a}:{a + 3}]")
        elif kind == "v_mfma_f64_4x4x4":
            a = 2 * (i % 8)
            out.append(f"v_mfma_f64_4x4x4_4b_f64 a[{a}:{a + 1}], v[{2 * (k % 8)}:{2 * (k % 8) + 1}], v[{16 + 2 * (j % 8)}:{17 + 2 * (j % 8)}], a[{a}:{a + 1}]")
        elif kind == "v_mfma_f64_16x16x4":
            a = 8 * (i % 4)
            out.append(f"v_mfma_f64_16x16x4_f64 a[{a}:{a + 7}], v[{2 * (k % 8)}:{2 * (k % 8) + 1}], v[{16 + 2 * (j % 8)}:{17 + 2 * (j % 8)}], a[{a}:{a + 7}]")
        elif kind == "v_mfma_f32_16x16x4":
            a = 4 * (i % 8)
            out.append(f"v_mfma_f32_16x16x4_f32 a[{a}:{a + 3}], v{k}, v{16 + j}, a[{a}:{a + 3}]")
        else:
            raise SystemExit(kind)
    return out


def mix_body(weights, n=256):
    """The classes interleaved in proportion to `weights` (largest-remainder schedule): the kernel's own mix."""
    tot = sum(weights.values())
    acc = {k: 0.0 for k in weights}
    seqs = {k: iter(body(k, 4 * n) if not k.startswith("ds_") else [x for x in body(k, 4 * n) if not x.startswith("s_wait")])
            for k in weights}
    out, lds = [], 0
    for _ in range(n):
        for k in weights:
            acc[k] += weights[k] / tot
        k = max(acc, key=acc.get)
        acc[k] -= 1.0
        out.append(next(seqs[k]))
        if k.startswith("ds_"):
            lds += 1
            if lds % 8 == 0:
                out.append("s_waitcnt lgkmcnt(0)")
    out.append("s_waitcnt lgkmcnt(0)")
    return out


SINGLE = ["v_add_f32", "v_sub_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mov_b32", "v_mov_b32_dpp", "v_add_f32_dpp",
          "v_cndmask_b32", "v_cndmask_b32_dpp_smov", "v_cndmask_b32_vcc_salu", "v_cndmask_b32_sgpr", "v_cndmask_b32_indep", "v_bfi_b32", "v_and_or_b32", "v_cndmask_b32_dpp", "v_log_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32", "ds_read_b64", "ds_read_b128", "ds_write_b32",
          "ds_write2_b32",
          "v_mfma_f32_4x4x1", "v_mfma_f32_16x16x4", "v_mfma_f64_4x4x4", "v_mfma_f64_16x16x4",
          "v_fma_f64", "v_add_f64", "v_mul_f64", "v_fma_f64_sgpr", "v_fmac_f64", "v_fmac_f64_dpp", "ds_read2_b64", "ds_read_b64_bcast"]


def kernel(name, lines, n_counted):
    asm = "\\n\\t".join(lines)
    return f'''
__global__ __launch_bounds__(1024) void k_{name}(unsigned long long *out, int iters) {{
  extern __shared__ float lds[];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  unsigned long long t0, t1;
  asm volatile(
      "v_mbcnt_lo_u32_b32 v32, -1, 0\\n\\tv_mbcnt_hi_u32_b32 v32, -1, v32\\n\\tv_lshlrev_b32 v32, 3, v32\\n\\t"
      "v_cmp_gt_u32 vcc, 3, v32\\n\\ts_mov_b64 s[22:23], 0x8001\\n\\tv_cndmask_b32 v35, 0, -1, vcc\\n\\t"
      "v_mov_b32 v36, 0\\n\\ts_mov_b32 s20, %[it]\\n\\t"
      "s_memtime %[t0]\\n\\ts_waitcnt lgkmcnt(0)\\n\\t"
      "L_{name}_%=:\\n\\t"
      "{asm}\\n\\t"
      "s_sub_u32 s20, s20, 1\\n\\ts_cmp_lg_u32 s20, 0\\n\\ts_cbranch_scc1 L_{name}_%=\\n\\t"
      "s_memtime %[t1]\\n\\ts_waitcnt lgkmcnt(0)\\n\\t"
      : [t0] "=&s"(t0), [t1] "=&s"(t1)
      : [it] "s"(iters)
      : "memory", "vcc", "scc", "s20", "s22", "s23", {", ".join(f'"v{i}"' for i in range(37))}, {", ".join(f'"a{i}"' for i in range(32))});
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}}
static const int n_{name} = {n_counted};
'''


def main():
    mix_path = os.path.join(HERE, "mfcc_wave_mix.json")
    mixes = {}
    if os.path.exists(mix_path):
        mixes = json.load(open(mix_path))
    src = ['// GENERATED by scripts/ubench/gen_mix_rate.py — do not edit', '#include <hip/hip_runtime.h>', '#include <cstdio>',
           '#include <vector>', '#include <algorithm>']
    tests = []
    for k in SINGLE:
        lines = body(k)
        src.append(kernel(k, lines, sum(1 for x in lines if not x.startswith(("s_waitcnt", "s_mov", "s_nop")))))
        tests.append(k)
    for name, w in mixes.items():
        lines = mix_body(w)
        src.append(kernel("mix_" + name, lines, sum(1 for x in lines if not x.startswith("s_waitcnt"))))
        tests.append("mix_" + name)
    src.append('''
template <class K> void run(const char *name, K kern, int n_instr) {
  const int iters = 4000;
  unsigned long long *out;
  (void)hipMalloc(&out, 256 * 16 * 8);
  for (int waves : {4, 8, 16}) {   // wavefronts per CU = 1, 2, 4 per SIMD; one workgroup per CU (96 KB of LDS each)
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(waves * 64), 96 * 1024, 0, out, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(waves * 64), 96 * 1024, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256 * waves);
    (void)hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double cyc = (double)h[h.size() / 2], n = (double)iters * n_instr;
    printf("%-28s waves/SIMD=%d  %7.2f memtime-ticks/instr/wave  %6.3f instr/tick/SIMD   (%.3f ms: %.2f nominal-2.4GHz cycles/instr/wave)\\n",
           name, waves / 4, cyc / n, n * (waves / 4.0) / cyc, ms, ms * 1e-3 * 2.4e9 / n);
  }
  (void)hipFree(out);
}
int main() {''')
    for t in tests:
        src.append(f'  (void)hipFuncSetAttribute((const void *)k_{t}, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);')
        src.append(f'  run("{t}", k_{t}, n_{t});')
    src.append('  return 0;\n}')
    open(os.path.join(HERE, "mix_rate.hip"), "w").write("\n".join(src) + "\n")
    print("wrote mix_rate.hip with", len(tests), "tests")


if __name__ == "__main__":
    main()
