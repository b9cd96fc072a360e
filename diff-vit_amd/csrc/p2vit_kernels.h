// internal launch interface between the C ABI (p2vit_capi.cpp) and the kernels (p2vit_gemm.hip, p2vit_ln.hip, p2vit_attn.hip, p2vit_misc.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/p2vit.h"

#define GBK_PAD 64   // K granularity of the GEMM tiles: weights/activations are padded to this

struct GemmArgs {
  const int8_t* A;      // activations [M][lda] int8
  int lda, M;
  const int8_t* W;      // weight codes [n_pad][K] int8, or (w4) packed int4 tiles [n_pad/128][K/64][128][32 B]
  int K, N;
  int w4;               // 1: W is the packed int4 layout (p2v_linear.packed4)
  const float* colscale;
  const float* bias;
  p2v_epilogue ep;
  void* out;
  int ldo;
  int8_t* out_codes;
  int tiles_n;          // filled by the launcher
#ifdef P2V_DIAG
  unsigned long long* stamps;   // diagnostic build only: per-workgroup cycle stamps (6 per workgroup) or null
#endif
};

// LayerNorm constants folded ahead of the launches (p2v_ln.pre, include/p2vit.h) instead of by every workgroup (ln_prepare).
// gm == nullptr: the kernel folds them itself.
typedef p2v_ln_pre LnPre;

struct LnArgs {
  const int8_t* x;
  long long row_stride;
  long long rows;
  int C;
  p2v_ln ln;
  int8_t* out;
  long long out_stride;
  int rows_per_half;    // filled by the launcher: consecutive rows per 32-lane half wave
  int force_generic;    // filled by the launcher (P2V_LN_GENERIC=1: A/B and parity runs of the generic chain)
  LnPre pre;            // optional, see above
};

struct AttnArgs {
  const int8_t* qkv;
  int B, N, H;
  p2v_attn at;
  int8_t* out;
  int8_t* probs_k;
  int pshift;           // filled by the launcher: score multiplier = 2^-pshift (>= 1) -> integer requant path; 0 = fp32 path
#ifdef P2V_DIAG
  unsigned long long* stamps;   // diagnostic build only: 16 cycle stamps per workgroup (wave 0) or null
#endif
};

struct WinAttnArgs {
  const int8_t* qkv;
  int B, T, H;          // images, tokens per image, heads
  p2v_winattn wa;
  int8_t* out;
  int8_t* probs_k;
};
int p2v_launch_patch_merge_gather(const int8_t* x, int B, int H, int W, int C, int8_t* out, hipStream_t st);
int p2v_launch_avgpool_quant(const int8_t* x, int B, int T, int C, float s_in, float inv_s_out, int8_t* out, hipStream_t st);
int p2v_launch_window_attention(const WinAttnArgs& a, hipStream_t st);

int p2v_launch_patchify(const float* img, int B, int C, int H, int W, int P, float inv_s, int8_t* out, int k_pad, hipStream_t st);
int p2v_launch_fill_cls(int8_t* x, int B, int T, int D, const int8_t* cls, hipStream_t st);
// fp32 image x fake-quantised conv weights (input_quant = False): EMBED epilogue, g.A unused, g.W unpacked int8 codes [n_pad][K]
int p2v_launch_embed_fp32(const float* img, int B, int C, int H, int W, int P, const GemmArgs& g, hipStream_t st);
int p2v_launch_gemm(int epi, const GemmArgs& g, hipStream_t st);
// table of the pre-folded RESID epilogue (p2v_epilogue.resid_tab): [ceil(N/128)][6][128] floats; flags: dev [2] preset to {1, 0}
int p2v_launch_resid_prefold(const p2v_linear& lin, const p2v_epilogue& ep, int N, float* tab, unsigned* flags, hipStream_t st);
int p2v_launch_layernorm(const LnArgs& a, hipStream_t st);
bool p2v_ln_gemm_supported(int epi, int C, int N, int table_cells);
int p2v_launch_ln_gemm(int epi, const LnArgs& a, const GemmArgs& g, hipStream_t st);   // -3: shape not fused
int p2v_launch_attention(const AttnArgs& a, int head_dim, hipStream_t st);
int p2v_launch_attention_stream(const AttnArgs& a, int head_dim, hipStream_t st);   // any token count up to P2V_MAX_TOKENS_STREAMED (p2vit_attn_stream.hip)
// tokens per image the resident attention kernel covers (K / V^T of a head in LDS: 3 * head_dim bytes per key); 0: head_dim not instantiated
inline int p2v_resident_tokens_of(int head_dim) {
  if (head_dim == 32 || head_dim == 48 || head_dim == 64 || head_dim == 80) return P2V_MAX_TOKENS;
  return head_dim == 96 ? 17 * 32 : (head_dim == 128 ? 12 * 32 : 0);
}
int p2v_launch_stream_probe(int kernels, int usec, int workgroups, int lds_bytes, hipStream_t st);     // trains of one-wave timed waits (p2vit_misc.hip)
int p2v_launch_fake_quant(const float* x, long long n, const float* scale, int n_scale, long long inner, int lo, int hi,
                          float* out, int8_t* codes, hipStream_t st);
int p2v_launch_gelu_quant(const float* y, long long n, float inv_s, int8_t* codes, unsigned long long* flags, int force_slow,
                          hipStream_t st);
int p2v_launch_gelu_sweep(unsigned first_bits, unsigned count, float* max_err, hipStream_t st);
// exact GELU->requant threshold table (see the GELU section of p2vit_device.h)
int p2v_launch_gelu_table_build(float inv_s, const p2v_gelu_tab& t, unsigned* scratch, hipStream_t st);
int p2v_launch_gelu_table_check(float inv_s, const p2v_gelu_tab& t, unsigned long long* mismatches, hipStream_t st);
