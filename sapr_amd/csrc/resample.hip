// Polyphase FIR sample-rate conversion on gfx950 — the resampling step of librosa.load(path)
// (mfcc_extract.py:12: every file is brought to 22 050 Hz before the MFCC chain).  librosa uses
// soxr_hq, whose filter is not reproducible here; this kernel implements scipy.signal.resample_poly's
// definition (zero-insertion by `up`, Kaiser-windowed low-pass designed on the host, decimation by
// `down`) so it can be checked against scipy on the CPU.  One lane per output sample, ~2*10*max(up,down)/up
// taps each; utterances are batched through offset tables.
#include "sapr_common.h"

namespace sapr {
namespace {

__global__ void resample_poly_kernel(const float *__restrict__ x, const int64_t *__restrict__ in_off,
                                     const int64_t *__restrict__ out_off, int64_t n_utts, int up, int down,
                                     const float *__restrict__ h, int n_taps, int n_pre_remove,
                                     float *__restrict__ y) {
  const int64_t u = blockIdx.y;
  if (u >= n_utts) return;
  const int64_t ib = in_off[u], ob = out_off[u];
  const int64_t n_in = in_off[u + 1] - ib, n_out = out_off[u + 1] - ob;
  for (int64_t n = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; n < n_out;
       n += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t m = (n + n_pre_remove) * down;  // position in the zero-stuffed, filtered stream
    int64_t j = m / up;                           // newest input sample that contributes
    int64_t k = m - j * up;                       // its tap; older samples use k + up, k + 2*up, ...
    if (j >= n_in) {
      const int64_t skip = j - (n_in - 1);
      j -= skip;
      k += skip * up;
    }
    double acc = 0.0;  // float64 accumulation like scipy's upfirdn on float64 input
    for (; j >= 0 && k < n_taps; --j, k += up) acc += static_cast<double>(h[k]) * static_cast<double>(x[ib + j]);
    y[ob + n] = static_cast<float>(acc);
  }
}

// 16-bit PCM -> float32 in [-1, 1): x / 32768 (exact: a power-of-two scale), the conversion the host WAV
// reader applies — done here so that host-resident audio crosses PCIe as 2 bytes per sample
__global__ void pcm16_to_f32_kernel(const int16_t *__restrict__ in, int64_t n, float *__restrict__ out) {
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    out[i] = static_cast<float>(in[i]) * (1.0f / 32768.0f);
}

}  // namespace
}  // namespace sapr

using namespace sapr;

extern "C" int sapr_pcm16_to_f32(const int16_t *pcm16, int64_t n_samples, float *out, void *stream) {
  SAPR_REQUIRE(n_samples >= 0, "bad size");
  if (n_samples == 0) return 0;
  SAPR_REQUIRE(pcm16 && out, "NULL pointer argument");
  const int64_t blocks = (n_samples + 255) / 256;
  SAPR_LAUNCH(pcm16_to_f32_kernel, dim3(static_cast<unsigned>(blocks > 65536 ? 65536 : blocks)), dim3(256), 0,
              as_stream(stream), pcm16, n_samples, out);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sapr_resample_poly(const float *x, const int64_t *in_offsets, const int64_t *out_offsets,
                                  int64_t n_utts, int64_t max_out, int32_t up, int32_t down, const float *taps,
                                  int32_t n_taps, int32_t n_pre_remove, float *y, void *stream) {
  SAPR_REQUIRE(n_utts >= 0 && up > 0 && down > 0 && n_taps > 0 && n_pre_remove >= 0 && max_out >= 0, "bad sizes");
  if (n_utts == 0 || max_out == 0) return 0;
  SAPR_REQUIRE(x && in_offsets && out_offsets && taps && y, "NULL pointer argument");
  SAPR_REQUIRE(n_utts <= 65535, "at most 65535 utterances per call");
  const unsigned bx = static_cast<unsigned>((max_out + 255) / 256 > 1024 ? 1024 : (max_out + 255) / 256);
  SAPR_LAUNCH(resample_poly_kernel, dim3(bx, static_cast<unsigned>(n_utts)), dim3(256), 0, as_stream(stream), x,
              in_offsets, out_offsets, n_utts, up, down, taps, n_taps, n_pre_remove, y);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}
