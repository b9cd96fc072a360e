// p2vit_capi.cpp -- extern "C" surface of libp2vit_hip.so (see include/p2vit.h).
//
// The plan object is the frozen integer state the reference keeps in Python attributes after
// model_close_calibrate(); model_quant() (test_quant.py:248-249): quantizer.scale / dic_scale / best_*
// lists.  p2v_forward walks the graph of VisionTransformer.forward_features/forward
// (models/vit_fquant.py:700-799) and enqueues 5 kernels per block on the caller's stream (LayerNorm+qkv, attention, proj+residual,
// LayerNorm+fc1+GELU, fc2+residual; 7 for models wider than the fused LayerNorm+GEMM kernel covers).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "p2vit_kernels.h"

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}
static int launch_rc(int rc, const char* what) {
  if (rc == 0) return 0;
  if (rc == -2) return fail(P2V_E_UNSUPPORTED, "%s: s_qkv_sq * inv_s_attn must be a power of two", what);
  if (rc < 0) return fail(P2V_E_UNSUPPORTED, "%s: no kernel instantiated for this shape", what);
  return fail(P2V_E_LAUNCH, "%s: %s", what, hipGetErrorString((hipError_t)rc));
}
static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
static bool is_pot(float v) {
  int ex;
  return v > 0.f && frexpf(v, &ex) == 0.5f;
}

// ---- constant checks shared by the per-operator entry points and the plan setters: a plan that would make a kernel leave its
//      exact range is refused when it is built, not when (or, through p2v_forward, never) an operator is called
// exp_int = z * 2^(32-q) with z = r (r + b) + c: the kernels keep z exactly in fp32 and the table in int64
static int check_lis_consts(const char* who, int x0_int, int b_int, int c_int) {
  if (x0_int >= 0) return fail(P2V_E_ARG, "%s: x0_int must be negative", who);
  if (c_int <= 0 || c_int >= (1 << 24) || b_int < 0 || b_int >= (1 << 23) || x0_int < -(1 << 12))
    return fail(P2V_E_UNSUPPORTED, "%s: log-int-softmax constants out of range (x0 %d, b %d, c %d): softmax input scale below 2^-11?", who, x0_int,
                b_int, c_int);
  return P2V_OK;
}
static int check_requant_scale(const char* who, float inv_s_out) {
  if (!(is_pot(inv_s_out) && inv_s_out >= 0x1p-40f && inv_s_out <= 0x1p40f))
    return fail(P2V_E_UNSUPPORTED, "%s: REQUANT 1/scale %g must be a power of two (it is folded into the column scales)", who, inv_s_out);
  return P2V_OK;
}
static int check_gelu_tab(const char* who, const p2v_gelu_tab& t) {
  if (t.table && (t.cells <= 0 || t.cells > 4096)) return fail(P2V_E_ARG, "%s: GELU table with %d cells", who, t.cells);
  return P2V_OK;
}
static int check_attn(const char* who, const p2v_attn& at) {
  const int rc = check_lis_consts(who, at.x0_int, at.b_int, at.c_int);
  if (rc != P2V_OK) return rc;
  // the kernel folds s_q1^2 / s_attn into qk_scale: exact only for a power of two (both are PoT scales in the reference)
  if (!is_pot(at.s_qkv_sq * at.inv_s_attn)) return fail(P2V_E_UNSUPPORTED, "%s: s_qkv_sq * inv_s_attn must be a power of two", who);
  if (!(at.qk_scale > 0.f) || !(at.av_mul > 0.f)) return fail(P2V_E_ARG, "%s: qk_scale and av_mul must be positive", who);
  // the probabilities enter the P.V product scaled by 2^-111 and av_mul carries the 2^111 (p2vit_attn.hip, lis_prob_pair): exact for a
  // power of two in this range (s_q1 / qact2 scale of two PoT activation scales)
  if (!is_pot(at.av_mul) || at.av_mul > 0x1p15f || at.av_mul < 0x1p-40f)
    return fail(P2V_E_UNSUPPORTED, "%s: av_mul %g must be a power of two in [2^-40, 2^15]", who, at.av_mul);
  return P2V_OK;
}

struct p2v_plan {
  p2v_model_desc d;
  int tokens, patches, k_patch, k_patch_pad, n_layers;
  std::vector<p2v_linear> lin[2];   // [bit index][layer]
  std::vector<char> lin_set[2];
  float inv_s_input;
  p2v_epilogue embed_epi;
  const int8_t* cls_codes;
  bool embed_set;
  std::vector<p2v_block> blocks;
  std::vector<char> block_set;
  p2v_ln final_ln;
  float head_inv_s, head_s;
  bool head_set;
  // LayerNorm constants folded at plan creation (p2v_ln.pre of the plan's copies of ln1 / ln2 / the final norm): the arrays live in one device
  // buffer per block (+ one for the final norm), owned by the plan
  std::vector<float*> ln_pre_buf;
  float* final_pre_buf = nullptr;
  std::vector<char> block_folded;
  // tables of the pre-folded RESID epilogue (p2v_resid_prefold): [block][proj = 0 / fc2 = 1][bit index]; null = the generic epilogue.  One device
  // buffer per block, owned by the plan
  std::vector<const float*> resid_tab;
  std::vector<float*> resid_buf;
  int device = -1;                  // the device that owns the folded constants (= the device of the caller's arrays)
  ~p2v_plan() {
    int prev = -1;
    const bool sw = device >= 0 && hipGetDevice(&prev) == hipSuccess && prev != device && hipSetDevice(device) == hipSuccess;
    for (float* b : ln_pre_buf)
      if (b) (void)hipFree(b);
    for (float* b : resid_buf)
      if (b) (void)hipFree(b);
    if (final_pre_buf) (void)hipFree(final_pre_buf);
    if (sw) (void)hipSetDevice(prev);
  }
};

static void build_resid_tables(p2v_plan* plan, int block);
static int bit_index(int bits) { return bits == 4 ? 0 : (bits == 8 ? 1 : -1); }

extern "C" {

extern int g_ln_generic;
extern int g_ln_rows;
extern int g_attn_waves;
extern int g_ln_gemm;
extern int g_ln_gemm_ver;
extern int g_gemm_tile;
extern int g_resid_pre;
extern int g_attn_stream;
extern int g_ln_pre;
// Tuning / A-B switches read once per process (first plan or first version query).  None of them changes results:
// P2V_LN_GENERIC forces the generic LayerNorm chain (bit-identical to the fast one, both are tested).
static void read_env_once() {
  static bool done = false;
  if (done) return;
  done = true;
  const char* e = getenv("P2V_ATTN_WAVES");
  if (e && atoi(e) >= 4 && atoi(e) <= 8) g_attn_waves = atoi(e);
  e = getenv("P2V_LN_GEMM");
  if (e) g_ln_gemm = atoi(e) != 0;
  e = getenv("P2V_LN_GEMM_V");
  if (e && atoi(e) >= 1 && atoi(e) <= 3) g_ln_gemm_ver = atoi(e);
  e = getenv("P2V_GEMM_TILE");
  if (e && (atoi(e) == 0 || atoi(e) == 128 || atoi(e) == 256)) g_gemm_tile = atoi(e);
  e = getenv("P2V_ATTN_STREAM");
  if (e) g_attn_stream = atoi(e) != 0;
  e = getenv("P2V_LN_PRE");
  if (e) g_ln_pre = atoi(e) != 0;
  e = getenv("P2V_RESID_PRE");
  if (e) g_resid_pre = atoi(e) != 0;
  e = getenv("P2V_LN_ROWS");
  if (e && atoi(e) >= 1 && atoi(e) <= 64) g_ln_rows = atoi(e);
  e = getenv("P2V_LN_GENERIC");
  if (e && atoi(e) == 1) g_ln_generic = 1;
}
int p2v_abi_version(void) { read_env_once(); return P2V_ABI_VERSION; }

int p2v_resident_tokens(int head_dim) { return p2v_resident_tokens_of(head_dim); }
int p2v_max_tokens(int head_dim) { return p2v_resident_tokens_of(head_dim) ? P2V_MAX_TOKENS_STREAMED : 0; }

int p2v_set_tuning(const char* name, int value) {
  if (!name) return fail(P2V_E_ARG, "p2v_set_tuning: null name");
  read_env_once();                      // an explicit setting wins over the environment
  if (!strcmp(name, "ln_gemm")) { g_ln_gemm = value != 0; return P2V_OK; }
  if (!strcmp(name, "ln_gemm_version") && value >= 1 && value <= 3) { g_ln_gemm_ver = value; return P2V_OK; }
  if (!strcmp(name, "ln_generic")) { g_ln_generic = value != 0; return P2V_OK; }
  if (!strcmp(name, "ln_rows") && value >= 1 && value <= 64) { g_ln_rows = value; return P2V_OK; }
  if (!strcmp(name, "attn_waves") && value >= 4 && value <= 8) { g_attn_waves = value; return P2V_OK; }
  if (!strcmp(name, "resid_pre")) { g_resid_pre = value != 0; return P2V_OK; }
  if (!strcmp(name, "ln_pre")) { g_ln_pre = value != 0; return P2V_OK; }
  if (!strcmp(name, "attn_stream")) { g_attn_stream = value != 0; return P2V_OK; }
  if (!strcmp(name, "gemm_tile") && (value == 0 || value == 128 || value == 256)) { g_gemm_tile = value; return P2V_OK; }
  return fail(P2V_E_ARG, "p2v_set_tuning: unknown switch or value out of range: %s = %d", name, value);
}
const char* p2v_last_error(void) { return g_err; }

int p2v_plan_create(const p2v_model_desc* desc, p2v_plan** out) {
  if (!desc || !out) return fail(P2V_E_ARG, "p2v_plan_create: null argument");
  read_env_once();
  if (desc->abi_version != P2V_ABI_VERSION) return fail(P2V_E_ARG, "ABI version %d != %d", desc->abi_version, P2V_ABI_VERSION);
  const p2v_model_desc& d = *desc;
  if (d.img_size <= 0 || d.patch_size <= 0 || d.img_size % d.patch_size) return fail(P2V_E_SHAPE, "img_size %% patch_size != 0");
  if (d.patch_size % 4) return fail(P2V_E_UNSUPPORTED, "patch_size must be a multiple of 4");
  if (d.embed_dim % d.num_heads) return fail(P2V_E_SHAPE, "embed_dim %% num_heads != 0");
  const int hd = d.embed_dim / d.num_heads;
  if (hd != 32 && hd != 48 && hd != 64 && hd != 80 && hd != 96 && hd != 128)
    return fail(P2V_E_UNSUPPORTED, "head_dim %d (32, 48, 64, 80, 96 and 128 are instantiated)", hd);
  // rows are read in 16-byte pieces and stored 16 channels at a time; widths that are not multiples of the 64-deep k-tile run through zero
  // weight columns (the tile's last pieces then belong to the next row of the workspace buffer: multiplied by zero)
  if (d.embed_dim % 16 || d.mlp_hidden % 16) return fail(P2V_E_UNSUPPORTED, "embed_dim and mlp_hidden must be multiples of 16");
  {   // the resident attention kernel keeps a query block's scores in registers and K / V^T of an image's head in LDS (p2v_resident_tokens); the
      // streaming kernel takes over beyond, up to 4096 tokens per image
    const int tokens = (d.img_size / d.patch_size) * (d.img_size / d.patch_size) + 1;
    if (tokens > P2V_MAX_TOKENS_STREAMED)
      return fail(P2V_E_UNSUPPORTED, "%d tokens per image: the attention kernels cover up to %d", tokens, P2V_MAX_TOKENS_STREAMED);
  }
  if (d.embed_dim > 2048) return fail(P2V_E_UNSUPPORTED, "embed_dim %d: the LayerNorm kernel covers up to 2048 channels", d.embed_dim);
  p2v_plan* p = new p2v_plan();
  p->d = d;
  p->patches = (d.img_size / d.patch_size) * (d.img_size / d.patch_size);
  p->tokens = p->patches + 1;
  p->k_patch = d.in_chans * d.patch_size * d.patch_size;
  p->k_patch_pad = round_up(p->k_patch, GBK_PAD);
  p->n_layers = 4 * d.depth + 2;
  for (int b = 0; b < 2; ++b) {
    p->lin[b].assign(p->n_layers, p2v_linear{nullptr, nullptr, nullptr, nullptr, 0});
    p->lin_set[b].assign(p->n_layers, 0);
  }
  p->blocks.resize(d.depth);
  p->block_set.assign(d.depth, 0);
  p->ln_pre_buf.assign(d.depth, nullptr);
  p->block_folded.assign(d.depth, 0);
  p->resid_tab.assign((size_t)d.depth * 4, nullptr);
  p->resid_buf.assign(d.depth, nullptr);
  p->embed_set = p->head_set = false;
  p->cls_codes = nullptr;
  *out = p;
  return P2V_OK;
}

void p2v_plan_destroy(p2v_plan* plan) { delete plan; }

int p2v_plan_set_linear(p2v_plan* plan, int layer, int bits, const p2v_linear* lin) {
  if (!plan || !lin) return fail(P2V_E_ARG, "p2v_plan_set_linear: null argument");
  const int bi = bit_index(bits);
  if (bi < 0) return fail(P2V_E_BITS, "bits %d not in {4, 8}", bits);
  if (layer < 0 || layer >= plan->n_layers) return fail(P2V_E_ARG, "layer %d out of range [0,%d)", layer, plan->n_layers);
  if (!lin->w_codes || !lin->colscale || !lin->bias) return fail(P2V_E_ARG, "p2v_plan_set_linear: null device pointer");
  if (lin->packed4 && bits != 4) return fail(P2V_E_BITS, "packed4 weights for a %d-bit layer", bits);
  plan->lin[bi][layer] = *lin;
  plan->lin_set[bi][layer] = 1;
  if (layer >= 1 && layer < plan->n_layers - 1 && ((layer - 1) & 1)) {         // proj / fc2 of block (layer - 1) / 4: their RESID tables depend on these constants
    const int block = (layer - 1) / 4;
    if (plan->block_set[block]) {
      char keep[sizeof g_err];
      memcpy(keep, g_err, sizeof keep);
      build_resid_tables(plan, block);        // unusable tables simply leave the generic RESID epilogue in place
      memcpy(g_err, keep, sizeof keep);
    }
  }
  return P2V_OK;
}

int p2v_plan_set_embed(p2v_plan* plan, float inv_s_input, const p2v_epilogue* e, const int8_t* cls_row_codes) {
  if (!plan || !e || !cls_row_codes) return fail(P2V_E_ARG, "p2v_plan_set_embed: null argument");
  plan->inv_s_input = inv_s_input;
  plan->embed_epi = *e;
  plan->embed_epi.patches = plan->patches;
  plan->cls_codes = cls_row_codes;
  plan->embed_set = true;
  return P2V_OK;
}

// Make the device that owns `ptr` current for the lifetime of the object (plan-time folds run hipMalloc / copies / small kernels next to the
// caller's arrays, whatever device the process has selected); ok() is false when the owner cannot be determined.
struct OwnerDevice {
  int prev = -1, own = -1;
  explicit OwnerDevice(const void* ptr) {
    hipPointerAttribute_t pa;
    if (hipGetDevice(&prev) != hipSuccess || hipPointerGetAttributes(&pa, ptr) != hipSuccess) {
      (void)hipGetLastError();
      return;
    }
    if (pa.type != hipMemoryTypeDevice && pa.type != hipMemoryTypeManaged) return;      // host memory (registered or not): not ours to launch on
    own = pa.device;
    if (own != prev && hipSetDevice(own) != hipSuccess) own = -1;
  }
  ~OwnerDevice() { if (own >= 0 && own != prev) (void)hipSetDevice(prev); }
  bool ok() const { return own >= 0; }
};

static size_t resid_tab_floats(int N) { return (size_t)((N + 127) / 128) * 6 * 128; }

// build one table on the current device; *usable = 0 when the pre-folded form is not provably the reference's for these constants
static int resid_prefold_impl(const p2v_linear& lin, const p2v_epilogue& ep, int N, float* tab, int* usable, hipStream_t st) {
  *usable = 0;
  unsigned* flags = nullptr;
  hipError_t e = hipMalloc(&flags, 2 * sizeof(unsigned));
  const unsigned init[2] = {1u, 0u};
  unsigned got[2] = {0u, 0u};
  if (e == hipSuccess) e = hipMemcpyAsync(flags, init, sizeof init, hipMemcpyHostToDevice, st);
  int rc = 0;
  if (e == hipSuccess) rc = p2v_launch_resid_prefold(lin, ep, N, tab, flags, st);
  if (e == hipSuccess && rc == 0) e = hipMemcpyAsync(got, flags, sizeof got, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess && rc == 0) e = hipStreamSynchronize(st);
  if (flags) (void)hipFree(flags);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(P2V_E_LAUNCH, "resid_prefold: %s", hipGetErrorString(e));
  }
  if (rc) return launch_rc(rc, "resid_prefold");
  *usable = got[0] ? 1 : 0;
  return P2V_OK;
}

// (Re)build the RESID tables of a block for every bit width whose weights are set; called from both setters, whichever comes last.
static void build_resid_tables(p2v_plan* plan, int block) {
  for (int j = 0; j < 4; ++j) plan->resid_tab[(size_t)block * 4 + j] = nullptr;
  if (!plan->block_set[block]) return;
  const p2v_block& blk = plan->blocks[block];
  OwnerDevice dev(blk.proj_epi.s_mid);
  if (!dev.ok() || (plan->device >= 0 && plan->device != dev.own)) return;
  const int D = plan->d.embed_dim;
  const size_t per = resid_tab_floats(D);
  if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); return; }      // the caller's uploads may still be in flight
  float* buf = plan->resid_buf[block];
  if (!buf && hipMalloc(&buf, 4 * per * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return; }
  plan->resid_buf[block] = buf;
  plan->device = dev.own;
  for (int which = 0; which < 2; ++which)
    for (int bi = 0; bi < 2; ++bi) {
      const int layer = 1 + 4 * block + (which ? 3 : 1);
      if (!plan->lin_set[bi][layer]) continue;
      int usable = 0;
      float* tab = buf + (size_t)(which * 2 + bi) * per;
      if (resid_prefold_impl(plan->lin[bi][layer], which ? blk.fc2_epi : blk.proj_epi, D, tab, &usable, nullptr) == P2V_OK && usable)
        plan->resid_tab[(size_t)block * 4 + which * 2 + bi] = tab;
    }
}

// The fold of ln_prepare (p2vit_ln.hip), once per LayerNorm instead of once per workgroup: the same fp32 products and the same tests on the
// host (this file is compiled without contraction too); `dev` receives gamma / out_scale and beta / out_scale, each padded with zeros to
// Cp = round_up(C, 256) channels.  The current device must be the one that owns the arrays.
static hipError_t fold_one_ln(const p2v_ln& l, int C, float* dev, p2v_ln_pre* out) {
  const int Cp = round_up(C, 256);                             // (a row group of 64 lanes covers 256 channels per chunk)
  std::vector<float> g(C), b(C), io(C), pm(C), host((size_t)2 * Cp, 0.f);
  hipError_t e = hipMemcpy(g.data(), l.gamma, C * sizeof(float), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(b.data(), l.beta, C * sizeof(float), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(io.data(), l.inv_out, C * sizeof(float), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(pm.data(), l.post_mul, C * sizeof(float), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return e;
  float* go = host.data();
  float* bo = go + Cp;
  int pot = 1, pm1 = 1;
  float gmin = 3.0e38f, gmax = 0.f, bmax = 0.f;
  for (int c = 0; c < C; ++c) {
    unsigned ib;
    memcpy(&ib, &io[c], 4);
    const int p2 = (int)((ib & 0x807FFFFFu) == 0u) & (int)((ib >> 23) - 32u <= 190u);    // +2^e, far from under/overflow
    go[c] = g[c] * io[c];
    bo[c] = b[c] * io[c];
    const float ga = fabsf(go[c]), ba = fabsf(bo[c]);
    const int gok = (int)(g[c] == 0.f) | ((int)(ga >= 1.0e-30f) & (int)(ga <= 1.0e30f));
    const int bok = (int)(b[c] == 0.f) | ((int)(ba >= 1.0e-30f) & (int)(ba <= 1.0e30f));
    pot &= p2 & gok & bok;
    pm1 &= (int)(pm[c] == 1.f);
    gmin = fminf(gmin, ga);
    gmax = fmaxf(gmax, ga);
    bmax = fmaxf(bmax, ba);
  }
  e = hipMemcpy(dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) *out = p2v_ln_pre{dev, dev + Cp, gmin, gmax, bmax, pot, pm1};
  return e;
}
static const p2v_ln_pre kNoPre = {nullptr, nullptr, 0.f, 0.f, 0.f, 0, 0};

// The plan's six LayerNorms of a block, folded on the device that OWNS the caller's arrays, whatever the process's current device is (a
// plan built for cuda:1 while cuda:0 is current).  Returns false, with the reason in p2v_last_error(), when the constants could not be
// folded: the block's kernels then fold per workgroup, with the same results.
static bool fold_ln_constants(p2v_plan* plan, int block) {
  const int C = plan->d.embed_dim, Cp = round_up(C, 256);
  p2v_block& blk = plan->blocks[block];
  auto ln_at = [&](int i) -> p2v_ln& { return i < 2 ? blk.ln1[i] : blk.ln2[(i - 2) >> 1][(i - 2) & 1]; };
  // stale state first: a second p2v_plan_set_block on this block must never leave the constants of the previous arrays behind
  for (int i = 0; i < 6; ++i) ln_at(i).pre = kNoPre;
  OwnerDevice own(blk.ln1[0].gamma);
  if (!own.ok()) {
    fail(P2V_OK, "p2v_plan_set_block: LayerNorm constants of block %d not folded at plan time (cannot locate the device of gamma)", block);
    return false;
  }
  if (plan->device >= 0 && plan->device != own.own) {
    fail(P2V_OK, "p2v_plan_set_block: block %d lives on device %d, earlier blocks on device %d: not folded at plan time", block, own.own, plan->device);
    return false;
  }
  hipError_t e = hipDeviceSynchronize();                        // the caller's uploads may still be in flight on another stream
  float* dev = plan->ln_pre_buf[block];
  if (e == hipSuccess && !dev) e = hipMalloc(&dev, (size_t)12 * Cp * sizeof(float));
  p2v_ln_pre pre[6];
  if (e == hipSuccess) {
    plan->ln_pre_buf[block] = dev;
    plan->device = own.own;
    for (int i = 0; i < 6 && e == hipSuccess; ++i) e = fold_one_ln(ln_at(i), C, dev + (size_t)(2 * i) * Cp, &pre[i]);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();                                    // do not leave the error for an unrelated CHECK_LAUNCH
    fail(P2V_OK, "p2v_plan_set_block: LayerNorm constants of block %d not folded at plan time (%s)", block, hipGetErrorString(e));
    return false;
  }
  for (int i = 0; i < 6; ++i) ln_at(i).pre = pre[i];
  return true;
}

int p2v_plan_set_block(p2v_plan* plan, int block, const p2v_block* blk) {
  if (!plan || !blk) return fail(P2V_E_ARG, "p2v_plan_set_block: null argument");
  if (block < 0 || block >= plan->d.depth) return fail(P2V_E_ARG, "block %d out of range", block);
  // p2v_forward launches the kernels directly: everything the per-operator entry points check is checked here, once
  int rc = check_attn("p2v_plan_set_block: attn", blk->attn);
  if (rc == P2V_OK) rc = check_gelu_tab("p2v_plan_set_block: gelu_fc1", blk->gelu_fc1);
  for (int b = 0; b < 2 && rc == P2V_OK; ++b) rc = check_requant_scale("p2v_plan_set_block: inv_s_qkv", blk->inv_s_qkv[b]);
  if (rc == P2V_OK && !(is_pot(blk->inv_s_fc1) && blk->inv_s_fc1 >= 0x1p-40f && blk->inv_s_fc1 <= 0x1p40f))
    rc = fail(P2V_E_UNSUPPORTED, "p2v_plan_set_block: 1/scale of mlp.qact1 %g must be a power of two", blk->inv_s_fc1);
  if (rc == P2V_OK && (!blk->proj_epi.s_mid || !blk->proj_epi.s_res || !blk->proj_epi.s_next || !blk->fc2_epi.s_mid || !blk->fc2_epi.s_res ||
                       !blk->fc2_epi.s_next))
    rc = fail(P2V_E_ARG, "p2v_plan_set_block: RESID epilogues need s_mid/s_res/s_next");
  for (int i = 0; i < 6 && rc == P2V_OK; ++i) {
    const p2v_ln& l = i < 2 ? blk->ln1[i] : blk->ln2[(i - 2) >> 1][(i - 2) & 1];
    if (!l.mask || !l.gamma || !l.beta || !l.inv_out || !l.post_mul) rc = fail(P2V_E_ARG, "p2v_plan_set_block: LayerNorm constants missing");
  }
  if (rc != P2V_OK) return rc;
  plan->blocks[block] = *blk;
  plan->block_set[block] = 1;
  // without the fold the kernels fold per workgroup, as for the per-operator calls: same results, so the call still succeeds; the reason
  // is left in p2v_last_error() and p2v_plan_block_prefolded() tells
  g_err[0] = 0;
  plan->block_folded[block] = fold_ln_constants(plan, block) ? 1 : 0;
  {
    char keep[sizeof g_err];
    memcpy(keep, g_err, sizeof keep);
    build_resid_tables(plan, block);        // unusable tables simply leave the generic RESID epilogue in place
    memcpy(g_err, keep, sizeof keep);
  }
  return P2V_OK;
}

int p2v_plan_block_prefolded(const p2v_plan* plan, int block) {
  if (!plan || block < 0 || block >= plan->d.depth) return fail(P2V_E_ARG, "p2v_plan_block_prefolded: bad argument");
  return plan->block_folded[block] ? 1 : 0;
}

int p2v_plan_resid_prefolded(const p2v_plan* plan, int block) {
  if (!plan || block < 0 || block >= plan->d.depth) return fail(P2V_E_ARG, "p2v_plan_resid_prefolded: bad argument");
  int m = 0;
  for (int j = 0; j < 4; ++j) m |= plan->resid_tab[(size_t)block * 4 + j] ? 1 << j : 0;
  return m;
}

int p2v_plan_set_head(p2v_plan* plan, const p2v_ln* final_ln, float inv_s_out, float s_out) {
  if (!plan || !final_ln) return fail(P2V_E_ARG, "p2v_plan_set_head: null argument");
  if (!final_ln->mask || !final_ln->gamma || !final_ln->beta || !final_ln->inv_out || !final_ln->post_mul)
    return fail(P2V_E_ARG, "p2v_plan_set_head: LayerNorm constants missing");
  if (!(inv_s_out > 0.f) || !(s_out > 0.f)) return fail(P2V_E_ARG, "p2v_plan_set_head: act_out scale must be positive");
  plan->final_ln = *final_ln;
  plan->final_ln.pre = kNoPre;
  {   // the final norm's constants, folded like a block's (best effort: the per-workgroup fold gives the same codes)
    OwnerDevice own(final_ln->gamma);
    const int C = plan->d.embed_dim;
    if (own.ok() && (plan->device < 0 || plan->device == own.own) && hipDeviceSynchronize() == hipSuccess) {
      if (!plan->final_pre_buf && hipMalloc(&plan->final_pre_buf, (size_t)2 * round_up(C, 256) * sizeof(float)) != hipSuccess) plan->final_pre_buf = nullptr;
      p2v_ln_pre pre;
      if (plan->final_pre_buf && fold_one_ln(*final_ln, C, plan->final_pre_buf, &pre) == hipSuccess) {
        plan->final_ln.pre = pre;
        plan->device = own.own;
      }
    }
    (void)hipGetLastError();
  }
  plan->head_inv_s = inv_s_out;
  plan->head_s = s_out;
  plan->head_set = true;
  return P2V_OK;
}

// workspace: [patches | x | ln | qkv | att | hid | cls], each 256-byte aligned
struct WsLayout {
  size_t patches, x, ln, qkv, att, hid, cls, total;
};
static WsLayout ws_layout(const p2v_plan* p, int batch) {
  auto al = [](size_t v) { return (v + 255) / 256 * 256; };
  const size_t M = (size_t)batch * p->tokens, D = p->d.embed_dim;
  WsLayout w;
  size_t off = 0;
  w.patches = off; off += al((size_t)batch * p->patches * p->k_patch_pad);
  w.x = off;       off += al(M * D);
  w.ln = off;      off += al(M * D);
  w.qkv = off;     off += al(M * 3 * D);
  w.att = off;     off += al(M * D);
  w.hid = off;     off += al(M * (size_t)p->d.mlp_hidden);
  w.cls = off;     off += al((size_t)batch * D);
  w.total = off + 256;     // a k-tile of the last row of the last buffer may reach up to 48 bytes past the row (widths not a multiple of 64)
  return w;
}

size_t p2v_workspace_bytes(const p2v_plan* plan, int batch) {
  if (!plan || batch <= 0) return 0;
  return ws_layout(plan, batch).total;
}

long long p2v_workspace_view(const p2v_plan* plan, int batch, const char* name) {
  if (!plan || !name || batch <= 0) return -1;
  WsLayout w = ws_layout(plan, batch);
  if (!strcmp(name, "patches")) return (long long)w.patches;
  if (!strcmp(name, "x")) return (long long)w.x;
  if (!strcmp(name, "ln")) return (long long)w.ln;
  if (!strcmp(name, "qkv")) return (long long)w.qkv;
  if (!strcmp(name, "att")) return (long long)w.att;
  if (!strcmp(name, "hid")) return (long long)w.hid;
  if (!strcmp(name, "cls")) return (long long)w.cls;
  return -1;
}

static int run_gemm(int epi, const int8_t* A, int lda, int M, int K, int N, const p2v_linear& lin, const p2v_epilogue& ep, void* out,
                    int ldo, int8_t* out_codes, hipStream_t st) {
  GemmArgs g;
  g.A = A; g.lda = lda; g.M = M; g.W = lin.w_codes; g.K = K; g.N = N; g.w4 = lin.packed4 ? 1 : 0;
  g.colscale = lin.colscale; g.bias = lin.bias; g.ep = ep; g.out = out; g.ldo = ldo; g.out_codes = out_codes; g.tiles_n = 0;
#ifdef P2V_DIAG
  g.stamps = nullptr;
#endif
  return launch_rc(p2v_launch_gemm(epi, g, st), "gemm_i8");
}

// QIntLayerNorm -> /cs -> qact0 -> QLinear -> (GELU) -> QAct in one launch (k_ln_gemm); a.out may be null
static int run_ln_gemm(int epi, const LnArgs& a, const p2v_linear& lin, const p2v_epilogue& ep, int N, int8_t* out, hipStream_t st) {
  GemmArgs g;
  if (!lin.w_frag) return fail(P2V_E_UNSUPPORTED, "ln_gemm: the layer has no fragment-order weights (p2v_linear.w_frag)");
  g.A = nullptr; g.lda = a.C; g.M = (int)a.rows; g.W = lin.w_frag; g.K = round_up(a.C, GBK_PAD); g.N = N; g.w4 = lin.packed4 ? 1 : 0;
  g.colscale = lin.colscale; g.bias = lin.bias; g.ep = ep; g.out = out; g.ldo = N; g.out_codes = nullptr; g.tiles_n = 0;
#ifdef P2V_DIAG
  g.stamps = nullptr;
#endif
  const int rc = p2v_launch_ln_gemm(epi, a, g, st);
  if (rc == -3) return fail(P2V_E_UNSUPPORTED, "ln_gemm: shape C=%d N=%d is not fused", a.C, N);
  return launch_rc(rc, "ln_gemm");
}

struct Prof {
  std::vector<hipEvent_t> ev;   // a pool created BEFORE the enqueue loop (an event created between two launches paces the host, and the
                                // interval it opens then measures the host, not the kernel); ev[i] is recorded before launch i, one more after the last
  std::vector<int> kind;        // P2V_K_* of launch i
  int used = 0;
  bool make_pool(int n) {
    ev.resize(n);
    for (int i = 0; i < n; ++i)
      if (hipEventCreate(&ev[i]) != hipSuccess) { ev.resize(i); return false; }
    return true;
  }
  ~Prof() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); }
};
static int prof_pool_size(const p2v_plan* p) { return p ? 7 * p->d.depth + 10 : 0; }

static int forward_impl(p2v_plan* p, const float* images, int batch, const int8_t* bit_config, int n_cfg, float* logits, void* workspace,
                        size_t workspace_bytes, int stop_after, void* stream, Prof* prof, float* const* qkv_tap = nullptr,
                        float* const* fc1_tap = nullptr) {
  if (!p || !images || !bit_config || !logits || !workspace) return fail(P2V_E_ARG, "p2v_forward: null argument");
  if (batch <= 0) return fail(P2V_E_SHAPE, "batch must be positive");
  if (n_cfg != p->n_layers) return fail(P2V_E_BITS, "bit_config has %d entries, model needs %d", n_cfg, p->n_layers);
  for (int i = 0; i < n_cfg; ++i) {
    const int bi = bit_index(bit_config[i]);
    if (bi < 0) return fail(P2V_E_BITS, "%d is not in list", (int)bit_config[i]);   // bit_pool.index, vit_fquant.py:282
    if (!p->lin_set[bi][i]) return fail(P2V_E_STATE, "layer %d has no %d-bit weights", i, (int)bit_config[i]);
  }
  if (!p->embed_set || !p->head_set) return fail(P2V_E_STATE, "plan incomplete (embed/head)");
  for (int i = 0; i < p->d.depth; ++i)
    if (!p->block_set[i]) return fail(P2V_E_STATE, "plan incomplete (block %d)", i);
  const WsLayout w = ws_layout(p, batch);
  if (workspace_bytes < w.total) return fail(P2V_E_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, w.total);
  hipStream_t st = (hipStream_t)stream;
  int8_t* ws = reinterpret_cast<int8_t*>(workspace);
  int8_t *bufP = ws + w.patches, *bufX = ws + w.x, *bufLN = ws + w.ln, *bufQKV = ws + w.qkv, *bufATT = ws + w.att,
         *bufHID = ws + w.hid, *bufCLS = ws + w.cls;
  const p2v_model_desc& d = p->d;
  const int D = d.embed_dim, T = p->tokens, M = batch * T, Hd = d.mlp_hidden, hd = D / d.num_heads;
  const int Dk = round_up(D, GBK_PAD), Hk = round_up(Hd, GBK_PAD);     // contraction depths in whole k-tiles (weights are zero-padded to them)
  int launched = 0, rc;
  const bool taps = stop_after >= 0;      // parity runs read the workspace buffers: the fused kernels then also write the LayerNorm codes
#define STEP(kind_, call)                             \
  do {                                                \
    if (stop_after >= 0 && launched >= stop_after) return P2V_OK; \
    if (prof) {                                       \
      if (prof->used + 1 >= (int)prof->ev.size()) return fail(P2V_E_LAUNCH, "profile: event pool exhausted"); \
      hipEventRecord(prof->ev[prof->used++], st);     \
      prof->kind.push_back(kind_);                    \
    }                                                 \
    rc = (call);                                      \
    if (rc) return rc;                                \
    launched += (kind_ == P2V_K_LN_GEMM_QKV || kind_ == P2V_K_LN_GEMM_FC1) ? 2 : 1;   /* a fused launch fills two slots of the stop_after numbering */ \
  } while (0)

  // qact_input + PatchEmbed + cls/pos/qact1                                 vit_fquant.py:705-733
  if (p->inv_s_input > 0.f) {
    STEP(P2V_K_PATCHIFY, launch_rc(p2v_launch_patchify(images, batch, d.in_chans, d.img_size, d.img_size, d.patch_size, p->inv_s_input, bufP,
                                       p->k_patch_pad, st), "quantize_patchify"));
    const p2v_linear& l = p->lin[bit_index(bit_config[0])][0];
    STEP(P2V_K_GEMM_EMBED, run_gemm(P2V_EPI_EMBED, bufP, p->k_patch_pad, batch * p->patches, p->k_patch_pad, D, l, p->embed_epi, bufX, D, nullptr, st));
  } else {
    // input_quant = False (vit_fquant.py:705, the vit_large factory :925): the fp32 image feeds the fake-quantised convolution
    const p2v_linear& l = p->lin[bit_index(bit_config[0])][0];
    if (l.packed4) return fail(P2V_E_UNSUPPORTED, "input_quant = False: the patch-embed weights must be unpacked codes (packed4 = 0)");
    GemmArgs g;
    g.A = nullptr; g.lda = 0; g.M = batch * p->patches; g.W = l.w_codes; g.K = p->k_patch_pad; g.N = D; g.w4 = 0;
    g.colscale = l.colscale; g.bias = l.bias; g.ep = p->embed_epi; g.out = bufX; g.ldo = D; g.out_codes = nullptr; g.tiles_n = 0;
#ifdef P2V_DIAG
    g.stamps = nullptr;
#endif
    STEP(P2V_K_GEMM_EMBED, launch_rc(p2v_launch_embed_fp32(images, batch, d.in_chans, d.img_size, d.img_size, d.patch_size, g, st), "embed_fp32"));
  }
  STEP(P2V_K_FILL_CLS, launch_rc(p2v_launch_fill_cls(bufX, batch, T, D, p->cls_codes, st), "fill_cls"));

  for (int i = 0; i < d.depth; ++i) {
    const p2v_block& b = p->blocks[i];
    const int8_t* bc = bit_config + 1 + 4 * i;
    const int bq = bit_index(bc[0]), bp = bit_index(bc[1]), b1 = bit_index(bc[2]), b2 = bit_index(bc[3]);
    // norm1 -> /channel_scale -> qact0                                     vit_fquant.py:431-434,284-289
    LnArgs ln{bufX, D, M, D, b.ln1[bq], bufLN, D};
    ln.pre = b.ln1[bq].pre;
    // qkv -> qact1                                                          vit_fquant.py:293,307
    p2v_epilogue e{};
    e.inv_s_out = b.inv_s_qkv[bq];
    e.tap_out = qkv_tap ? qkv_tap[i] : nullptr;
    if (p->lin[bq][1 + 4 * i].w_frag && p2v_ln_gemm_supported(P2V_EPI_REQUANT, D, 3 * D, 0)) {              // one launch: the LayerNorm output stays in LDS
      if (!taps) ln.out = nullptr;
      STEP(P2V_K_LN_GEMM_QKV, run_ln_gemm(P2V_EPI_REQUANT, ln, p->lin[bq][1 + 4 * i], e, 3 * D, bufQKV, st));
    } else {
      STEP(P2V_K_LAYERNORM, launch_rc(p2v_launch_layernorm(ln, st), "int_layernorm"));
      STEP(P2V_K_GEMM_QKV, run_gemm(P2V_EPI_REQUANT, bufLN, D, M, Dk, 3 * D, p->lin[bq][1 + 4 * i], e, bufQKV, 3 * D, nullptr, st));
    }
    // scores -> qact_attn1 -> log-int-softmax -> @v -> qact2                vit_fquant.py:309-326
    AttnArgs at{bufQKV, batch, T, d.num_heads, b.attn, bufATT, nullptr};
    STEP(P2V_K_ATTENTION, launch_rc(p2v_launch_attention(at, hd, st), "lis_attention"));
    // proj -> qact3 -> + x -> Block.qact2                                   vit_fquant.py:334-338,431
    p2v_epilogue ep = b.proj_epi;
    ep.residual = bufX;
    ep.resid_tab = p->resid_tab[(size_t)i * 4 + bp];
    STEP(P2V_K_GEMM_PROJ, run_gemm(P2V_EPI_RESID, bufATT, D, M, Dk, D, p->lin[bp][2 + 4 * i], ep, bufX, D, nullptr, st));
    // norm2 (attention's channel scale!) -> /mlp.channel_scale -> mlp.qact0 vit_fquant.py:464, layers_quant.py:305-311
    LnArgs ln2{bufX, D, M, D, b.ln2[bq][b1], bufLN, D};
    ln2.pre = b.ln2[bq][b1].pre;
    // fc1 -> GELU -> qact1                                                  layers_quant.py:316,331-333
    p2v_epilogue e1{};
    e1.inv_s_out = b.inv_s_fc1;
    e1.gelu = b.gelu_fc1;
    e1.tap_out = fc1_tap ? fc1_tap[i] : nullptr;
    if (p->lin[b1][3 + 4 * i].w_frag && p2v_ln_gemm_supported(P2V_EPI_GELU, D, Hd, e1.gelu.table ? e1.gelu.cells : 0)) {
      if (!taps) ln2.out = nullptr;
      STEP(P2V_K_LN_GEMM_FC1, run_ln_gemm(P2V_EPI_GELU, ln2, p->lin[b1][3 + 4 * i], e1, Hd, bufHID, st));
    } else {
      STEP(P2V_K_LAYERNORM, launch_rc(p2v_launch_layernorm(ln2, st), "int_layernorm"));
      STEP(P2V_K_GEMM_FC1, run_gemm(P2V_EPI_GELU, bufLN, D, M, Dk, Hd, p->lin[b1][3 + 4 * i], e1, bufHID, Hd, nullptr, st));
    }
    // fc2 -> qact2 -> + x -> Block.qact4                                    layers_quant.py:342-346, vit_fquant.py:468
    p2v_epilogue e2 = b.fc2_epi;
    e2.residual = bufX;
    e2.resid_tab = p->resid_tab[(size_t)i * 4 + 2 + b2];
    STEP(P2V_K_GEMM_FC2, run_gemm(P2V_EPI_RESID, bufHID, Hd, M, Hk, D, p->lin[b2][4 + 4 * i], e2, bufX, D, nullptr, st));
  }
  // norm over the cls rows only ([:,0]) -> qact2 -> head -> act_out         vit_fquant.py:766-796
  LnArgs lf{bufX, (long long)T * D, batch, D, p->final_ln, bufCLS, D};
  lf.pre = p->final_ln.pre;
  STEP(P2V_K_LAYERNORM, launch_rc(p2v_launch_layernorm(lf, st), "int_layernorm"));
  p2v_epilogue eh{};
  eh.inv_s_out = p->head_inv_s;
  eh.s_out = p->head_s;
  STEP(P2V_K_GEMM_HEAD, run_gemm(P2V_EPI_HEAD, bufCLS, D, batch, Dk, d.num_classes, p->lin[bit_index(bit_config[n_cfg - 1])][n_cfg - 1], eh, logits,
                d.num_classes, nullptr, st));
#undef STEP
  if (prof) {
    if (prof->used + 2 > (int)prof->ev.size()) return fail(P2V_E_LAUNCH, "profile: event pool exhausted");
    hipEventRecord(prof->ev[prof->used++], st);
    hipEventRecord(prof->ev[prof->used++], st);      // an empty interval: the cost of the event pair itself (P2V_K_EVENT_GAP)
    prof->kind.push_back(P2V_K_EVENT_GAP);
  }
  return P2V_OK;
}

int p2v_forward(p2v_plan* p, const float* images, int batch, const int8_t* bit_config, int n_cfg, float* logits, void* workspace,
                size_t workspace_bytes, int stop_after, void* stream) {
  return forward_impl(p, images, batch, bit_config, n_cfg, logits, workspace, workspace_bytes, stop_after, stream, nullptr);
}

int p2v_forward_taps(p2v_plan* p, const float* images, int batch, const int8_t* bit_config, int n_cfg, float* logits, void* workspace,
                     size_t workspace_bytes, float* const* qkv_out, float* const* fc1_out, void* stream) {
  return forward_impl(p, images, batch, bit_config, n_cfg, logits, workspace, workspace_bytes, -1, stream, nullptr, qkv_out, fc1_out);
}

static int prof_collect(Prof& prof, float* ms_out, int32_t* kind_out, int max_launches) {
  hipEventSynchronize(prof.ev[prof.used - 1]);
  int n = (int)prof.kind.size();
  n = n < max_launches ? n : max_launches;          // never more than the caller's arrays hold
  for (int i = 0; i < n; ++i) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, prof.ev[i], prof.ev[i + 1]);
    ms_out[i] = ms;
    kind_out[i] = prof.kind[i];
  }
  return n;
}

int p2v_forward_profile(p2v_plan* p, const float* images, int batch, const int8_t* bit_config, int n_cfg, float* logits, void* workspace,
                        size_t workspace_bytes, void* stream, float* ms_out, int32_t* kind_out, int max_launches) {
  if (!p || !ms_out || !kind_out || max_launches <= 0) return fail(P2V_E_ARG, "p2v_forward_profile: null argument");
  Prof prof;
  if (!prof.make_pool(prof_pool_size(p))) return fail(P2V_E_LAUNCH, "hipEventCreate failed");
  const int rc = forward_impl(p, images, batch, bit_config, n_cfg, logits, workspace, workspace_bytes, -1, stream, &prof);
  return rc == P2V_OK ? prof_collect(prof, ms_out, kind_out, max_launches) : rc;
}

int p2v_forward_profile_begin(p2v_plan* p, const float* images, int batch, const int8_t* bit_config, int n_cfg, float* logits, void* workspace,
                              size_t workspace_bytes, void* stream, void** token) {
  if (!p || !token) return fail(P2V_E_ARG, "p2v_forward_profile_begin: null argument");
  *token = nullptr;
  Prof* prof = new Prof();
  if (!prof->make_pool(prof_pool_size(p))) {
    delete prof;
    return fail(P2V_E_LAUNCH, "hipEventCreate failed");
  }
  const int rc = forward_impl(p, images, batch, bit_config, n_cfg, logits, workspace, workspace_bytes, -1, stream, prof);
  if (rc != P2V_OK) {
    delete prof;
    return rc;
  }
  *token = prof;
  return P2V_OK;
}

int p2v_forward_profile_end(void* token, float* ms_out, int32_t* kind_out, int max_launches) {
  if (!token) return fail(P2V_E_ARG, "p2v_forward_profile_end: null argument");
  Prof* prof = reinterpret_cast<Prof*>(token);
  int n = 0;
  if (ms_out && kind_out && max_launches > 0) n = prof_collect(*prof, ms_out, kind_out, max_launches);
  else hipEventSynchronize(prof->ev[prof->used - 1]);      // ms_out == NULL: just release the token (error paths of the caller)
  delete prof;
  return n;
}

// ---- per-operator entry points ------------------------------------------------------------------------
int p2v_quantize_patchify(const float* img, int batch, int chans, int height, int width, int patch, float inv_s, int8_t* out,
                          int k_pad, void* stream) {
  if (!img || !out) return fail(P2V_E_ARG, "p2v_quantize_patchify: null argument");
  if (patch <= 0 || patch % 4 || height % patch || width % patch) return fail(P2V_E_SHAPE, "image %dx%d not divisible into %d-patches", height, width, patch);
  if (k_pad % 4 || k_pad < chans * patch * patch) return fail(P2V_E_ARG, "k_pad %d too small / unaligned", k_pad);
  return launch_rc(p2v_launch_patchify(img, batch, chans, height, width, patch, inv_s, out, k_pad, (hipStream_t)stream), "quantize_patchify");
}

int p2v_gemm_i8(int kind, const int8_t* A, int lda, int M, int K, int N, const p2v_linear* lin, const p2v_epilogue* epi, void* out,
                int ldo, int8_t* out_codes, void* stream) {
  if (!A || !lin || !epi || !out) return fail(P2V_E_ARG, "p2v_gemm_i8: null argument");
  if (kind < P2V_EPI_REQUANT || kind > P2V_EPI_HEAD) return fail(P2V_E_ARG, "unknown epilogue %d", kind);
  if (M <= 0 || N <= 0 || K <= 0 || K % GBK_PAD) return fail(P2V_E_SHAPE, "K=%d must be a positive multiple of %d", K, GBK_PAD);
  if (kind != P2V_EPI_HEAD && (N % 16 || ldo % 16)) return fail(P2V_E_UNSUPPORTED, "N and ldo must be multiples of 16 for int8 outputs");
  if (lda % 16) return fail(P2V_E_UNSUPPORTED, "lda must be a multiple of 16");
  if (lda <= 0 || ldo < N) return fail(P2V_E_SHAPE, "p2v_gemm_i8: lda=%d must be positive and ldo=%d must cover the %d outputs of a row", lda, ldo, N);
  if (kind == P2V_EPI_REQUANT && check_requant_scale("p2v_gemm_i8", epi->inv_s_out) != P2V_OK) return P2V_E_UNSUPPORTED;
  if (kind == P2V_EPI_GELU && check_gelu_tab("p2v_gemm_i8", epi->gelu) != P2V_OK) return P2V_E_ARG;
  if (kind == P2V_EPI_RESID && (!epi->s_mid || !epi->s_res || !epi->s_next || !epi->residual)) return fail(P2V_E_ARG, "RESID epilogue needs s_mid/s_res/s_next/residual");
  if (kind == P2V_EPI_EMBED && (!epi->s_next || !epi->pos_deq || epi->patches <= 0)) return fail(P2V_E_ARG, "EMBED epilogue needs s_next/pos_deq/patches");
  return run_gemm(kind, A, lda, M, K, N, *lin, *epi, out, ldo, out_codes, (hipStream_t)stream);
}

size_t p2v_ln_prefold_bytes(int C) { return C > 0 ? (size_t)2 * round_up(C, 256) * sizeof(float) : 0; }

int p2v_ln_prefold(p2v_ln* ln, int C, float* buf, size_t buf_bytes) {
  if (!ln || !buf) return fail(P2V_E_ARG, "p2v_ln_prefold: null argument");
  ln->pre = kNoPre;
  if (C <= 0 || C % 4 || C > 2048) return fail(P2V_E_SHAPE, "p2v_ln_prefold: C must be a positive multiple of 4 up to 2048");
  if (!ln->gamma || !ln->beta || !ln->inv_out || !ln->post_mul) return fail(P2V_E_ARG, "p2v_ln_prefold: LayerNorm constants missing");
  if (buf_bytes < p2v_ln_prefold_bytes(C)) return fail(P2V_E_WORKSPACE, "p2v_ln_prefold: buffer %zu < %zu bytes", buf_bytes, p2v_ln_prefold_bytes(C));
  OwnerDevice own(buf);
  if (!own.ok()) return fail(P2V_E_ARG, "p2v_ln_prefold: buf is not a device pointer");
  hipError_t e = hipDeviceSynchronize();                        // the caller's uploads may still be in flight
  p2v_ln_pre pre;
  if (e == hipSuccess) e = fold_one_ln(*ln, C, buf, &pre);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(P2V_E_LAUNCH, "p2v_ln_prefold: %s", hipGetErrorString(e));
  }
  ln->pre = pre;
  return P2V_OK;
}

size_t p2v_resid_prefold_bytes(int N) { return N > 0 ? resid_tab_floats(N) * sizeof(float) : 0; }

int p2v_resid_prefold(const p2v_linear* lin, const p2v_epilogue* epi, int N, float* tab, size_t tab_bytes, int* usable, void* stream) {
  if (!lin || !epi || !tab || !usable) return fail(P2V_E_ARG, "p2v_resid_prefold: null argument");
  *usable = 0;
  if (N <= 0) return fail(P2V_E_SHAPE, "p2v_resid_prefold: N must be positive");
  if (!lin->colscale || !lin->bias || !epi->s_mid || !epi->s_res || !epi->s_next) return fail(P2V_E_ARG, "p2v_resid_prefold: needs colscale / bias / s_mid / s_res / s_next");
  if (tab_bytes < p2v_resid_prefold_bytes(N)) return fail(P2V_E_WORKSPACE, "p2v_resid_prefold: table %zu < %zu bytes", tab_bytes, p2v_resid_prefold_bytes(N));
  OwnerDevice dev(tab);
  if (!dev.ok()) return fail(P2V_E_ARG, "p2v_resid_prefold: tab is not a device pointer");
  return resid_prefold_impl(*lin, *epi, N, tab, usable, (hipStream_t)stream);
}

int p2v_int_layernorm(const int8_t* x, long long row_stride, int rows, int C, const p2v_ln* ln, int8_t* out, long long out_stride,
                      void* stream) {
  if (!x || !ln || !out) return fail(P2V_E_ARG, "p2v_int_layernorm: null argument");
  if (C % 4 || row_stride % 4 || out_stride % 4) return fail(P2V_E_UNSUPPORTED, "C and strides must be multiples of 4");
  if (rows <= 0) return fail(P2V_E_SHAPE, "rows must be positive");
  if (C <= 0 || row_stride < 0 || out_stride < C) return fail(P2V_E_SHAPE, "p2v_int_layernorm: C=%d must be positive, row_stride non-negative, out_stride >= C", C);
  LnArgs a{x, row_stride, rows, C, *ln, out, out_stride};
  a.pre = ln->pre;                       // optional: constants folded ahead of the launch (p2v_ln_prefold)
  return launch_rc(p2v_launch_layernorm(a, (hipStream_t)stream), "int_layernorm");
}

int p2v_ln_gemm_i8(int kind, const int8_t* x, long long row_stride, int M, int C, const p2v_ln* ln, int N, const p2v_linear* lin,
                   const p2v_epilogue* epi, int8_t* out, int ldo, int8_t* ln_out, void* stream) {
  if (!x || !ln || !lin || !epi || !out) return fail(P2V_E_ARG, "p2v_ln_gemm_i8: null argument");
  if (kind != P2V_EPI_REQUANT && kind != P2V_EPI_GELU) return fail(P2V_E_ARG, "p2v_ln_gemm_i8: epilogue %d (REQUANT and GELU are fused)", kind);
  if (M <= 0 || C <= 0 || N <= 0) return fail(P2V_E_SHAPE, "p2v_ln_gemm_i8: bad shape");
  if (C % 4 || row_stride % 4 || N % 16 || ldo != N) return fail(P2V_E_UNSUPPORTED, "p2v_ln_gemm_i8: C, row_stride multiples of 4; N multiple of 16; ldo == N");
  if (row_stride < 0) return fail(P2V_E_SHAPE, "p2v_ln_gemm_i8: negative row_stride");
  if (kind == P2V_EPI_REQUANT && check_requant_scale("p2v_ln_gemm_i8", epi->inv_s_out) != P2V_OK) return P2V_E_UNSUPPORTED;
  if (kind == P2V_EPI_GELU && check_gelu_tab("p2v_ln_gemm_i8", epi->gelu) != P2V_OK) return P2V_E_ARG;
  read_env_once();
  LnArgs a{x, row_stride, M, C, *ln, ln_out, C};
  a.pre = ln->pre;
  return run_ln_gemm(kind, a, *lin, *epi, N, out, (hipStream_t)stream);
}

int p2v_ln_gemm_fusable(int kind, int C, int N, int cells) {
  read_env_once();
  return (C > 0 && N > 0 && cells >= 0 && cells <= 4096 && p2v_ln_gemm_supported(kind, C, N, cells)) ? 1 : 0;
}

int p2v_lis_attention(const int8_t* qkv, int batch, int tokens, int heads, int head_dim, const p2v_attn* at, int8_t* out,
                      int8_t* probs_k, void* stream) {
  if (!qkv || !at || !out) return fail(P2V_E_ARG, "p2v_lis_attention: null argument");
  if (batch <= 0 || tokens <= 0 || heads <= 0) return fail(P2V_E_SHAPE, "bad attention shape");
  {
    const int rc = check_attn("p2v_lis_attention", *at);
    if (rc != P2V_OK) return rc;
  }
  AttnArgs a{qkv, batch, tokens, heads, *at, out, probs_k};
  return launch_rc(p2v_launch_attention(a, head_dim, (hipStream_t)stream), "lis_attention");
}

int p2v_patch_merge_gather(const int8_t* x, int batch, int H, int W, int C, int8_t* out, void* stream) {
  if (!x || !out) return fail(P2V_E_ARG, "p2v_patch_merge_gather: null argument");
  if (batch <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return fail(P2V_E_SHAPE, "patch merge: H and W must be positive and even");
  if (C <= 0 || C % 16) return fail(P2V_E_UNSUPPORTED, "patch merge: C must be a multiple of 16");
  return launch_rc(p2v_launch_patch_merge_gather(x, batch, H, W, C, out, (hipStream_t)stream), "patch_merge_gather");
}

int p2v_avgpool_quant(const int8_t* x, int batch, int tokens, int C, float s_in, float inv_s_out, int8_t* out, void* stream) {
  if (!x || !out) return fail(P2V_E_ARG, "p2v_avgpool_quant: null argument");
  if (batch <= 0 || tokens <= 0 || tokens > 8192) return fail(P2V_E_SHAPE, "avgpool: bad shape");
  if (C <= 0 || C % 4) return fail(P2V_E_UNSUPPORTED, "avgpool: C must be a multiple of 4");
  return launch_rc(p2v_launch_avgpool_quant(x, batch, tokens, C, s_in, inv_s_out, out, (hipStream_t)stream), "avgpool_quant");
}

int p2v_window_attention(const int8_t* qkv, int batch, int tokens_per_image, int heads, int head_dim, const p2v_winattn* wa,
                         int8_t* out, int8_t* probs_k, void* stream) {
  if (!qkv || !wa || !out || !wa->table_codes || !wa->win_index) return fail(P2V_E_ARG, "p2v_window_attention: null argument");
  if (batch <= 0 || tokens_per_image <= 0 || heads <= 0) return fail(P2V_E_SHAPE, "bad window attention shape");
  if (head_dim != 32) return fail(P2V_E_UNSUPPORTED, "window attention: head_dim must be 32");
  if (wa->ws < 1 || wa->ws > 8 || wa->n_windows < 1 || wa->ws * wa->ws * wa->n_windows > tokens_per_image)
    return fail(P2V_E_SHAPE, "window attention: window size must be 1..8 and windows must fit the token count");
  {
    const int rc = check_lis_consts("p2v_window_attention", wa->x0_int, wa->b_int, wa->c_int);
    if (rc != P2V_OK) return rc;
  }
  const float pots[5] = {wa->s_q1, wa->s_attn, wa->s_table, wa->s_q2, wa->s_q3};
  for (float s : pots) {
    int ex;
    if (!(s > 0.f) || frexpf(s, &ex) != 0.5f) return fail(P2V_E_UNSUPPORTED, "window attention: activation scales must be powers of two");
  }
  if (wa->qkv_stride % 16 || wa->out_stride % 4 || (wa->qkv_stride && wa->qkv_stride < 3 * heads * head_dim) ||
      (wa->out_stride && wa->out_stride < heads * head_dim))
    return fail(P2V_E_SHAPE, "window attention: row strides must cover a row and be 16 / 4 byte aligned");
  if (wa->s_q2 > 1.0f) return fail(P2V_E_UNSUPPORTED, "window attention: qact2 scale above 1 (the -100 mask would not be an integer)");
  if (wa->s_q1 / wa->s_q3 > 0x1p15f || wa->s_q1 / wa->s_q3 < 0x1p-40f)
    return fail(P2V_E_UNSUPPORTED, "window attention: qact1 scale / qact3 scale must lie in [2^-40, 2^15]");
  WinAttnArgs a{qkv, batch, tokens_per_image, heads, *wa, out, probs_k};
  return launch_rc(p2v_launch_window_attention(a, (hipStream_t)stream), "window_attention");
}

static int run_one_op(const p2v_op& o, void* stream) {
  switch (o.kind) {
    case P2V_OP_PATCHIFY:
      return p2v_quantize_patchify((const float*)o.in, o.i0, o.i1, o.i2, o.i3, o.i4, o.f0, (int8_t*)o.out, o.i5, stream);
    case P2V_OP_GEMM:
      return p2v_gemm_i8(o.epi, (const int8_t*)o.in, o.lda, o.M, o.K, o.N, &o.lin, &o.ep, o.out, o.ldo, nullptr, stream);
    case P2V_OP_LAYERNORM:
      return p2v_int_layernorm((const int8_t*)o.in, o.lda, o.M, o.N, &o.ln, (int8_t*)o.out, o.ldo, stream);
    case P2V_OP_WINATTN:
      return p2v_window_attention((const int8_t*)o.in, o.i0, o.i1, o.i2, o.i3, &o.wa, (int8_t*)o.out, nullptr, stream);
    case P2V_OP_MERGE:
      return p2v_patch_merge_gather((const int8_t*)o.in, o.i0, o.i1, o.i2, o.i3, (int8_t*)o.out, stream);
    case P2V_OP_AVGPOOL:
      return p2v_avgpool_quant((const int8_t*)o.in, o.i0, o.i1, o.i2, o.f0, o.f1, (int8_t*)o.out, stream);
    case P2V_OP_LN_GEMM:
      return p2v_ln_gemm_i8(o.epi, (const int8_t*)o.in, o.lda, o.M, o.K, &o.ln, o.N, &o.lin, &o.ep, (int8_t*)o.out, o.ldo, nullptr, stream);
    default:
      return fail(P2V_E_ARG, "p2v_run_ops: unknown op kind %d", o.kind);
  }
}

int p2v_run_ops(const p2v_op* ops, int n_ops, void* stream) {
  if (!ops || n_ops < 0) return fail(P2V_E_ARG, "p2v_run_ops: null argument");
  for (int i = 0; i < n_ops; ++i) {
    const int rc = run_one_op(ops[i], stream);
    if (rc != P2V_OK) return rc;
  }
  return P2V_OK;
}

int p2v_run_ops_profile(const p2v_op* ops, int n_ops, void* stream, float* ms) {
  if (!ops || !ms || n_ops < 0) return fail(P2V_E_ARG, "p2v_run_ops_profile: null argument");
  hipStream_t st = (hipStream_t)stream;
  std::vector<hipEvent_t> ev(n_ops + 1);
  for (auto& e : ev) hipEventCreate(&e);
  int rc = P2V_OK;
  hipEventRecord(ev[0], st);
  int done = 0;
  for (; done < n_ops; ++done) {
    rc = run_one_op(ops[done], stream);
    if (rc != P2V_OK) break;
    hipEventRecord(ev[done + 1], st);
  }
  hipStreamSynchronize(st);
  for (int i = 0; i < done; ++i) hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]);
  for (auto& e : ev) hipEventDestroy(e);
  return rc;
}

int p2v_fake_quant_f32(const float* x, long long n, const float* scale, int n_scale, long long inner, int lo, int hi, float* out,
                       int8_t* codes, void* stream) {
  if (!x || !scale || (!out && !codes)) return fail(P2V_E_ARG, "p2v_fake_quant_f32: null argument");
  if (n_scale < 1 || inner < 1) return fail(P2V_E_ARG, "n_scale and inner must be >= 1");
  if (n == 0) return P2V_OK;
  return launch_rc(p2v_launch_fake_quant(x, n, scale, n_scale, inner, lo, hi, out, codes, (hipStream_t)stream), "fake_quant");
}

int p2v_gelu_quant_f32(const float* y, long long n, float inv_s, int8_t* codes, unsigned long long* flags, int force_slow,
                       void* stream) {
  if (!y || !codes) return fail(P2V_E_ARG, "p2v_gelu_quant_f32: null argument");
  if (n == 0) return P2V_OK;
  return launch_rc(p2v_launch_gelu_quant(y, n, inv_s, codes, flags, force_slow, (hipStream_t)stream), "gelu_quant");
}

int p2v_stream_probe(void* stream, int kernels, int usec, int workgroups, int lds_bytes) {
  if (kernels < 1 || kernels > 64 || usec < 1 || usec > 1000 || workgroups < 1 || workgroups > 4096 || lds_bytes < 0 || lds_bytes > 65536)
    return fail(P2V_E_ARG, "p2v_stream_probe: 1..64 kernels of 1..1000 us, 1..4096 workgroups, 0..64 KB of LDS");
  return launch_rc(p2v_launch_stream_probe(kernels, usec, workgroups, lds_bytes, (hipStream_t)stream), "stream_probe");
}

#ifdef P2V_DIAG
/* diagnostic build only (make diag): device buffer receiving 6 x uint64 per workgroup of the next tiled GEMM launches */
extern unsigned long long* g_gemm_stamps;
void p2v_debug_set_gemm_stamps(void* dev) { g_gemm_stamps = (unsigned long long*)dev; }
#endif

int p2v_gelu_table_plan(float inv_s, p2v_gelu_tab* t) {
  if (!t) return fail(P2V_E_ARG, "p2v_gelu_table_plan: null argument");
  if (!is_pot(inv_s) || inv_s < 1.0f || inv_s > 4096.0f) return fail(P2V_E_UNSUPPORTED, "gelu table: 1/scale %g is not a power of two in [1, 4096]", inv_s);
  const double s = 1.0 / (double)inv_s, k = 2.0 * (double)inv_s;          // cells of width s/2
  // below y_lo every code is 0: |gelu(y)| = 0.5 |y| erfc(|y|/sqrt 2) falls monotonically left of the minimum at -0.7518
  double y_lo = -0.75;
  while (y_lo > -40.0 && 0.5 * -y_lo * erfc(-y_lo * 0.70710678118654752440) >= 0.25 * s) y_lo -= 0.5 * s;
  const long long i0 = (long long)floor((y_lo - s) * k);
  double y_hi = 127.5 * s;                                                  // gelu(y) < y: the code saturates at 127 a little later
  while (y_hi < 1.0e4 && 0.5 * y_hi * erfc(-y_hi * 0.70710678118654752440) < 127.75 * s) y_hi += 0.5 * s;
  const long long i1 = (long long)ceil(y_hi * k) + 2;                       // from here on every code is 127
  const long long cells = i1 - i0;
  if (cells > 4096) return fail(P2V_E_UNSUPPORTED, "gelu table: %lld cells for 1/scale %g (limit 4096)", cells, inv_s);
  t->k = (float)k;
  t->off = (float)(-i0);
  t->cells = (int32_t)cells;
  return P2V_OK;
}

size_t p2v_gelu_table_scratch_bytes(int cells) { return cells > 0 ? ((size_t)4 * cells + 4) * 4 : 0; }

int p2v_gelu_table_build(float inv_s, const p2v_gelu_tab* t, void* scratch, size_t scratch_bytes, void* stream) {
  if (!t || !t->table || !scratch) return fail(P2V_E_ARG, "p2v_gelu_table_build: null argument");
  p2v_gelu_tab want;
  const int rc = p2v_gelu_table_plan(inv_s, &want);
  if (rc != P2V_OK) return rc;
  if (want.k != t->k || want.off != t->off || want.cells != t->cells) return fail(P2V_E_ARG, "p2v_gelu_table_build: descriptor does not come from p2v_gelu_table_plan");
  if (scratch_bytes < p2v_gelu_table_scratch_bytes(t->cells)) return fail(P2V_E_WORKSPACE, "gelu table scratch %zu < %zu bytes", scratch_bytes, p2v_gelu_table_scratch_bytes(t->cells));
  hipStream_t st = (hipStream_t)stream;
  const int lrc = launch_rc(p2v_launch_gelu_table_build(inv_s, *t, (unsigned*)scratch, st), "gelu_table_build");
  if (lrc != P2V_OK) return lrc;
  unsigned status = 0;
  hipError_t e = hipMemcpyAsync(&status, (const unsigned*)scratch + 4 * t->cells, 4, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return fail(P2V_E_LAUNCH, "gelu_table_build: %s", hipGetErrorString(e));
  if (status) return fail(P2V_E_UNSUPPORTED, "gelu table for 1/scale %g: %s", inv_s, (status & 1) ? "a cell holds two thresholds" : "a cell is never hit");
  return P2V_OK;
}

int p2v_gelu_table_check(float inv_s, const p2v_gelu_tab* t, unsigned long long* mismatches, void* stream) {
  if (!t || !t->table || !mismatches) return fail(P2V_E_ARG, "p2v_gelu_table_check: null argument");
  if (t->cells <= 0) return fail(P2V_E_ARG, "p2v_gelu_table_check: empty table");
  return launch_rc(p2v_launch_gelu_table_check(inv_s, *t, mismatches, (hipStream_t)stream), "gelu_table_check");
}

int p2v_gelu_err_sweep(unsigned first_bits, unsigned count, float* max_err, void* stream) {
  if (!max_err) return fail(P2V_E_ARG, "p2v_gelu_err_sweep: null argument");
  return launch_rc(p2v_launch_gelu_sweep(first_bits, count, max_err, (hipStream_t)stream), "gelu_sweep");
}

}  // extern "C"
