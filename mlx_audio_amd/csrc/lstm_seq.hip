// Unidirectional LSTM over a whole sequence for hidden sizes that do not fit one CU (EnCodec: two layers of 512, gfx950).
//
// Replaces the reference's LSTM (codec/models/encodec/encodec.py:89-167): x @ Wx^T + b for every time step at once (the caller does that as ONE GEMM),
// then per time step  h @ Wh^T  and a custom Metal kernel for the gates (i, f, g, o chunks of the 4H pre-activations; its own sigmoid
// 1 / (1 + exp(-|x|)) mirrored for x < 0, precise tanh).  The persistent bidirectional kernel of lstm.hip keeps Wh of BOTH directions in the
// registers of one CU, which ends at H = 256; at H = 512 Wh is 2 MB.  Here the recurrence is a native loop of two launches per step:
//   mi355_gemv: pre[b, :] = h[b, :] @ Wh^T + xproj[b, t, :]   (Wh row-major 16-bit, 2 MB: L2 / Infinity-Cache resident after the first step; the
//               x-projection row rides in as the residual operand, 1..64 sequences per step)
//   lstm_gates_kernel: c = f * c + i * g,  h = o * tanh(c),  h also stored as row t of the output.
// T x 2 dependent launches (~3 us each): 750 steps of a 10 s clip cost ~5 ms per layer regardless of the batch.
// Round 5: ONE launch per step when the caller hands Wh with its rows ordered by hidden unit (gate_interleaved: row 4 j + g) -- a 16-row tile of the
// step's GEMM then holds all four gates of four units and the gates / cell update / stores are the GEMM's own epilogue (gemm_rows.hip, LSTM mode);
// h ping-pongs between two buffers because every workgroup of a step reads the whole previous h.
#include <string.h>
#include "common.h"

namespace {

__device__ __forceinline__ float lstm_sigmoid(float x) {   // encodec.py:94-98
  const float y = 1.0f / (1.0f + expf(-fabsf(x)));
  return x < 0.f ? 1.0f - y : y;
}

__global__ __launch_bounds__(256) void lstm_gates_kernel(const float* __restrict__ pre, int ldp, float* __restrict__ c, float* __restrict__ h, float* __restrict__ out,
                                                        int64_t out_bstride, int H, int B) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * H) return;
  const int b = idx / H, j = idx - b * H;
  const float* p = pre + (int64_t)b * ldp + j;
  const float i = lstm_sigmoid(p[0]), f = lstm_sigmoid(p[H]), g = tanhf(p[2 * H]), o = lstm_sigmoid(p[3 * H]);
  const float cn = f * c[idx] + i * g;
  const float hn = o * tanhf(cn);
  c[idx] = cn;
  h[idx] = hn;
  out[(int64_t)b * out_bstride + j] = hn;
}

}  // namespace

int mi355_gemm_rows_lstm_step(const mi355_gemv_args& a, const float* xproj, int64_t xproj_bstride, float* c, float* out, int64_t out_bstride, hipStream_t st);   // gemm_rows.hip

extern "C" int mi355_lstm_seq(const mi355_lstm_seq_args* ap, void* stream) {
  MI355_REQUIRE(ap && ap->xproj && ap->wh && ap->h && ap->c && ap->pre && ap->out, "lstm_seq: null tensor");
  const mi355_lstm_seq_args a = *ap;
  MI355_REQUIRE(a.B >= 1 && a.B <= 64 && a.T >= 1 && a.H >= 8 && a.H % 8 == 0, "lstm_seq: 1..64 sequences, H a multiple of 8 (got B %d, H %d)", a.B, a.H);
  MI355_REQUIRE(a.B <= 8 || a.H % 64 == 0, "lstm_seq: more than 8 sequences need H %% 64 == 0");
  MI355_REQUIRE(a.ld_xproj >= 4 * a.H && a.ld_out >= a.H && a.xproj_bstride >= (int64_t)a.T * a.ld_xproj && a.out_bstride >= (int64_t)a.T * a.ld_out,
                "lstm_seq: bad strides");
  hipStream_t st = (hipStream_t)stream;
  if (a.gate_interleaved) {
    MI355_REQUIRE(a.h2 && a.H % 64 == 0 && ((uintptr_t)a.wh) % 16 == 0 && ((uintptr_t)a.h) % 16 == 0 && ((uintptr_t)a.h2) % 16 == 0,
                  "lstm_seq: the one-launch step needs the h2 scratch, H %% 64 == 0 and 16-byte aligned wh / h / h2 (H = %d)", a.H);
    MI355_REQUIRE(a.wdtype == MI355_W_BF16 || a.wdtype == MI355_W_F16, "lstm_seq: wdtype must be MI355_W_BF16 or MI355_W_F16");
    float* hin = a.h;
    float* hout = a.h2;
    for (int t = 0; t < a.T; ++t) {
      mi355_gemv_args g;
      memset(&g, 0, sizeof(g));
      g.x = hin; g.ldx = a.H; g.M = a.B; g.K = a.H; g.w = a.wh; g.ldw = a.H; g.wdtype = a.wdtype; g.N = 4 * a.H; g.out_scale = 1.f; g.y = hout; g.ldy = a.H;
      const int rc = mi355_gemm_rows_lstm_step(g, a.xproj + (int64_t)t * a.ld_xproj, a.xproj_bstride, a.c, a.out + (int64_t)t * a.ld_out, a.out_bstride, st);
      if (rc) return rc;
      float* tmp = hin; hin = hout; hout = tmp;
    }
    if (hin != a.h) {   // an odd number of steps left the final state in the scratch
      hipError_t e = hipMemcpyAsync(a.h, hin, sizeof(float) * (size_t)a.B * a.H, hipMemcpyDeviceToDevice, st);
      MI355_REQUIRE(e == hipSuccess, "lstm_seq: copy of the final state failed: %s", hipGetErrorString(e));
    }
    return MI355_OK;
  }
  for (int t = 0; t < a.T; ++t) {
    mi355_gemv_args g;
    memset(&g, 0, sizeof(g));
    g.x = a.h; g.ldx = a.H; g.M = a.B; g.K = a.H; g.w = a.wh; g.ldw = a.H; g.wdtype = a.wdtype; g.N = 4 * a.H;
    g.res = a.xproj + (int64_t)t * a.ld_xproj; g.ldr = (int32_t)a.xproj_bstride; g.out_scale = 1.f; g.y = a.pre; g.ldy = 4 * a.H;
    int rc = mi355_gemv(&g, stream);
    if (rc) return rc;
    MI355_CLEAR_ERROR();
    hipLaunchKernelGGL(lstm_gates_kernel, dim3((a.B * a.H + 255) / 256), dim3(256), 0, st, a.pre, 4 * a.H, a.c, a.h, a.out + (int64_t)t * a.ld_out, a.out_bstride, a.H, a.B);
    MI355_LAUNCH_CHECK("lstm_seq(gates)");
  }
  return MI355_OK;
}
