/*
 * p2vit.h -- C ABI of the MI355X-native PoT-PTQ quantized ViT forward engine (libp2vit_hip.so).
 *
 * Drop-in boundary.  The reference (LeSN-Lab/diff-ViT) has no native code: its boundary is the Python
 * nn.Module surface of models/ptq/layers.py plus VisionTransformer.forward (models/vit_fquant.py:780).
 * This header is what a ctypes/cffi binding inside that surface calls when `.quant` is on; every entry
 * point names the reference function(s) it replaces.  Signatures carry only PODs, device pointers and a
 * hipStream_t (passed as void*): no torch types.
 *
 * Conventions
 *   - every pointer marked "dev" is a device (HBM) pointer borrowed for the duration of the call (plan
 *     setters: for the lifetime of the plan); the library never allocates or frees device memory and
 *     never synchronises; work is enqueued on `stream`.
 *   - return value: 0 = ok, negative = error (P2V_E_*); p2v_last_error() gives the message of the last
 *     failing call on the calling thread.  The Python shim maps P2V_E_BITS -> ValueError/KeyError (as
 *     bit_pool.index / BIT_TYPE_DICT lookups do, vit_fquant.py:282, layers.py:174), P2V_E_SHAPE ->
 *     AssertionError (layers_quant.py:437), P2V_E_UNSUPPORTED -> NotImplementedError (quantizer/base.py:28).
 *   - activations between kernels are int8 codes, row-major [rows][channels]; "rows" = batch*tokens.
 *   - all scales named *_pot are exact powers of two (the reference's PoT observers, minmax.py:247-251);
 *     per-channel PTF scales (ptf.py:51,133) are arbitrary fp32 and are divided by, as the reference does.
 *
 * Limits of what is instantiated (everything else returns P2V_E_UNSUPPORTED, at plan creation where the geometry is known):
 *   - ViT attention: head_dim 32, 48, 64, 80, 96 or 128 (round 4; before: 32 / 64); up to P2V_MAX_TOKENS_STREAMED = 4096 tokens per image.  The
 *     fast kernel keeps K / V^T of an image's head in LDS: P2V_MAX_TOKENS = 608 tokens (224^2 / 16 = 197, 384^2 / 16 = 577, ...), 544 at head_dim 96,
 *     384 at head_dim 128 (p2v_resident_tokens); beyond, a streaming kernel re-reads K / V per query block (round 4: slow, exact, so that no
 *     geometry of the reference's VisionTransformer is refused; 448^2 / 16 = 785, 512^2 / 16 = 1025);
 *     Swin window attention: head_dim 32, windows up to 8 x 8;
 *   - embed_dim and MLP width of a plan: multiples of 16 (round 4; before: 64) - the contractions walk 64-deep k-tiles through zero weight
 *     columns (p2v_linear: k_pad = round_up(K, 64)); the per-operator GEMM entry points take K in whole k-tiles;
 *   - LayerNorm: up to 2048 channels, PTF input masks (in_scale / min in_scale) in {1, 2, 4, 8};
 *   - LayerNorm output scale: p2v_ln.inv_out is MULTIPLIED by where the reference divides by the scale - identical for the power-of-two
 *     scales of this path; for any other scale pass p2v_ln.out_scale too and the kernel divides (exact, ABI 3);
 *   - REQUANT epilogues fold 1/scale into the column constants: it must be a power of two;
 *   - log-int-softmax constants: c_int < 2^24 (qact_attn1 scale >= 2^-11).
 */
#ifndef P2VIT_H
#define P2VIT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P2V_ABI_VERSION 5
#define P2V_MAX_TOKENS 608            /* tokens per image of the RESIDENT ViT attention kernel (K / V^T of a head in LDS; 19 pairs of 32 keys) */
#define P2V_MAX_TOKENS_STREAMED 4096  /* tokens per image of the streaming attention kernel that takes over beyond the resident one's limit (round 4) */
/* tokens per image a plan / p2v_lis_attention accepts at this head dimension (P2V_MAX_TOKENS_STREAMED; 0: head_dim not instantiated), and how many of
 * them the resident (fast) kernel covers: 608 up to head_dim 80, 544 at 96, 384 at 128 - launches beyond run the streaming kernel */
int p2v_max_tokens(int head_dim);
int p2v_resident_tokens(int head_dim);

enum {
  P2V_OK = 0,
  P2V_E_ARG = -1,         /* null pointer / bad size                                  */
  P2V_E_BITS = -2,        /* bit_config entry not in {4,8} or wrong length             */
  P2V_E_SHAPE = -3,       /* image / tensor shape does not match the plan              */
  P2V_E_UNSUPPORTED = -4, /* configuration the kernels are not instantiated for        */
  P2V_E_WORKSPACE = -5,   /* workspace too small                                       */
  P2V_E_LAUNCH = -6,      /* hipLaunch failure (message carries hipGetErrorString)     */
  P2V_E_STATE = -7        /* plan incomplete (a setter was not called)                 */
};

/* GEMM epilogues: what follows F.linear / F.conv2d in the reference graph. */
enum {
  P2V_EPI_REQUANT = 0, /* QLinear -> QAct (PoT): attn.qkv->qact1 (vit_fquant.py:293,307)                      */
  P2V_EPI_GELU = 1,    /* QLinear -> nn.GELU -> QAct (PoT): mlp.fc1->act->qact1 (layers_quant.py:316,331-333) */
  P2V_EPI_RESID = 2,   /* QLinear -> QAct(PTF) -> + residual -> QAct(PTF): proj->qact3->+x->Block.qact2
                          (vit_fquant.py:334-338,431) and fc2->qact2->+x->Block.qact4 (layers_quant.py:342-346,
                          vit_fquant.py:468)                                                                   */
  P2V_EPI_EMBED = 3,   /* QConv2d -> PatchEmbed.qact -> qact_embed -> + qact_pos(pos) -> qact1(PTF)
                          (layers_quant.py:467-491, vit_fquant.py:715-733)                                     */
  P2V_EPI_HEAD = 4     /* head -> act_out, fp32 logits on the int8 grid (vit_fquant.py:792-796)                */
};

typedef struct p2v_model_desc {
  int32_t abi_version; /* P2V_ABI_VERSION */
  int32_t img_size, patch_size, in_chans;
  int32_t embed_dim, depth, num_heads, mlp_hidden, num_classes;
} p2v_model_desc;

/* One fake-quantised weight matrix for one bit width (QLinear/QConv2d.forward, layers.py:82-88,173-178;
 * UniformQuantizer.quant, uniform.py:50-88).  packed4 == 0: codes one per byte ([-8,7] for 4-bit), row
 * major [n_pad][k_pad], n_pad = round_up(N,128), k_pad = round_up(K,64), zero padded.
 * packed4 == 1 (4-bit codes only): two codes per byte, stored as the LDS images of the GEMM's weight tiles --
 *   [n_pad/128 column tiles][k_pad/64 k-tiles][128 rows][32 bytes]; the 32 bytes of row r hold its 64 k-values as four 8-byte chunks
 *   c (k = 16c .. 16c+15) at chunk position c ^ ((r >> 3) & 3) (bank swizzle); inside a chunk byte j of dword 0 is
 *   code[j] | code[4+j] << 4 and byte j of dword 1 is code[8+j] | code[12+j] << 4 (low nibble first, two's complement nibbles).
 *   Half the bytes, one contiguous 4 KB block per tile; gfx950 has no int4 MFMA, the kernels widen the nibbles to int8 in registers.
 * colscale[n] = s_x * s_w[n]  (activation scale times per-tensor (int8) or per-out-channel (int4) weight
 * scale; both powers of two, so the product is exact).  bias is the un-quantised fp32 bias. */
typedef struct p2v_linear {
  const int8_t* w_codes; /* dev */
  const float* colscale; /* dev [n_pad] */
  const float* bias;     /* dev [n_pad] */
  const int8_t* w_frag;  /* dev, optional (NULL): the same codes in MFMA-fragment order for p2v_ln_gemm_i8 --
                          * [n_pad/128 column tiles][4 waves][k_pad/32 k-steps][64 lanes][16 bytes], lane = 32*h + r holding
                          * W[128*tile + 32*wave + r][32*kstep + 16*h .. +16): the A operand of v_mfma_i32_32x32x32_i8 as one
                          * coalesced 1 KB load per wave.  packed4 == 0: one code per byte (16 bytes per lane).  packed4 == 1 (ABI 4):
                          * the lane's 16 codes in 8 bytes, byte j of dword 0 = code[j] | code[4+j] << 4, of dword 1 =
                          * code[8+j] | code[12+j] << 4 (the chunk format of the packed tiles): half the weight bytes */
  int32_t packed4;       /* 1: w_codes is the packed int4 tile layout described above */
} p2v_linear;

/* QIntLayerNorm.forward mode 'int' (layers.py:255-289) fused with the division by the SmoothQuant
 * channel scale and the QAct that follows it (vit_fquant.py:284-289, layers_quant.py:305-311).
 *   x_q = code * mask[c];  mask[c] = round(in_scale[c] / s1), s1 = min_c in_scale[c]
 *   out = clamp(rne(LN_int(x_q) * post_mul[c]), -128, 127)
 * inv_out[c] = 1 / (out_quantizer.scale * out_quantizer_scale[c])  (a power of two in P2-ViT: multiplying by it IS the reference's
 *              division by the scale)
 * out_scale[c] = the scale itself, optional (NULL): needed only when it is NOT a power of two - the kernel then divides by it
 *              (IEEE) exactly where the reference does (layers.py:279-286); with NULL it multiplies by inv_out, which for such a
 *              scale lands the 8-bit multiplier M one step away on about 1e-5 of the elements
 * post_mul[c] = out_scale[c] / next_channel_scale[c] / next_act_scale  (power of two). */
/* Constants of a LayerNorm folded ahead of its launches (round 4, ABI 5; optional: gm == NULL and every kernel folds per workgroup, with the
 * same codes): gamma * inv_out and beta * inv_out, each padded with zeros to round_up(C, 256) channels, their extreme magnitudes and the two
 * tests of the fast chain (1 / out_scale a power of two everywhere; post_mul == 1 everywhere).  Filled by p2v_ln_prefold; the frozen plan of
 * p2v_forward folds its own copies and ignores what the caller passes here. */
typedef struct p2v_ln_pre {
  const float* gm;        /* dev [round_up(C, 256)] */
  const float* bt;        /* dev [round_up(C, 256)] */
  float gmin, gmax, bmax;
  int32_t pot, pm_one;
} p2v_ln_pre;

typedef struct p2v_ln {
  float s1;
  const float* mask;      /* dev [C] */
  const float* gamma;     /* dev [C] */
  const float* beta;      /* dev [C] */
  const float* inv_out;   /* dev [C] */
  const float* post_mul;  /* dev [C] */
  const float* out_scale; /* dev [C] or NULL (ABI 3) */
  p2v_ln_pre pre;         /* optional (ABI 5), see above */
} p2v_ln;
/* Fold the constants of *ln for C channels into buf (dev, p2v_ln_prefold_bytes(C) bytes, owned by the caller, must outlive the launches) and
 * set ln->pre.  Reads the arrays back to the host: synchronises the device; call it after the arrays have been written. */
size_t p2v_ln_prefold_bytes(int C);
int p2v_ln_prefold(p2v_ln* ln, int C, float* buf, size_t buf_bytes);

/* (q @ k^T) * scale -> qact_attn1 -> QIntSoftmax (log-int-softmax, uint4) -> @ v -> qact2
 * (vit_fquant.py:309-326, layers.py:323-376). */
typedef struct p2v_attn {
  float s_qkv_sq;   /* s_q1^2                                      */
  float qk_scale;   /* head_dim^-0.5 (vit_fquant.py:74)            */
  float inv_s_attn; /* 1 / qact_attn1 scale (pot)                  */
  float av_mul;     /* s_q1 / qact2 scale (pot)                    */
  int32_t x0_int, b_int, c_int; /* I-BERT exp polynomial constants (layers.py:334-351) for sf = qact_attn1 scale */
} p2v_attn;

/* nn.GELU -> QAct(PoT) (layers_quant.py:331-333) as an exact threshold table built by p2v_gelu_table_build:
 *   u = y * k (exact: k is a power of two) of the fp32 pre-activation y;  cell i = clamp(floor(u) + off, 0, cells-1);
 *   entry = { float thr * k; uint32 lo | hi << 8 };  code = (u >= thr * k) ? (int8)hi : (int8)lo.
 *   (Round 4: the epilogues form u straight from the accumulator, fma(acc, colscale * k, bias * k) = RN(acc * colscale + bias) * k, and never
 *   y itself; before, the cell was floor(fma(y, k, off)) and the entry held thr.  A table must come from THIS library's builder.)
 *   table == NULL selects the arithmetic evaluation (A&S erfc + fp64 fallback). */
typedef struct p2v_gelu_tab {
  const void* table; /* dev [cells] 8-byte entries, or NULL */
  float k, off;      /* k = 2 / s_out (a power of two: y*k is exact), off = -floor(y_lo * k), an integer      */
  int32_t cells;
} p2v_gelu_tab;

/* Epilogue parameters; which members are read depends on the epilogue kind. */
typedef struct p2v_epilogue {
  float inv_s_out;       /* REQUANT/GELU/HEAD: 1/scale of the following QAct (pot)           */
  float s_out;           /* HEAD: act_out scale                                               */
  const float* s_mid;    /* dev [N] RESID: PTF scale of qact3 / mlp.qact2                    */
  const float* s_res;    /* dev [N] RESID: scale of the residual-stream codes being added     */
  const float* s_next;   /* dev [N] RESID/EMBED: PTF scale of the QAct that ends the stage    */
  const int8_t* residual;/* dev [M][N] RESID: residual-stream codes (may alias the output)    */
  /* EMBED only */
  float inv_s_pe;        /* 1 / PatchEmbed.qact scale                                         */
  float pe_to_embed;     /* PatchEmbed.qact scale / qact_embed scale                          */
  float s_embed;         /* qact_embed scale                                                  */
  const float* pos_deq;  /* dev [tokens][N]  qact_pos(pos_embed), dequantised                 */
  int32_t patches;       /* patches per image; output row = b*(patches+1) + 1 + p             */
  p2v_gelu_tab gelu;     /* GELU only: threshold table for inv_s_out (optional)              */
  float* tap_out;        /* REQUANT/GELU, optional: dev fp32 [M][N] receives the layer output BEFORE the
                            following QAct / GELU, acc*colscale + bias -- what the reference keeps as
                            Attention.qkv_output (vit_fquant.py:301) / Mlp.fc1_output (layers_quant.py:326) */
  const float* resid_tab;/* RESID, optional (ABI 5): dev table written by p2v_resid_prefold for THIS (weights, epilogue) pair, or NULL.
                            With it the launch runs the pre-folded epilogue (same codes, ~4 VALU instructions per output less); the
                            frozen plan of p2v_forward builds and owns its tables itself */
} p2v_epilogue;

/* RESID epilogue constants folded ahead of the launches (round 4).  The reference computes, per output channel n (vit_fquant.py:334-338,431;
 * layers_quant.py:342-346, vit_fquant.py:468),
 *     q3 = clamp(round((acc * colscale[n] + bias[n]) / s_mid[n]));   q = clamp(round((res * s_res[n] + q3 * s_mid[n]) / s_next[n]))
 * with two IEEE divisions by non-power-of-two PTF scales.  The table holds, per column tile of 128 channels, six arrays of 128 floats:
 *     colscale * fl(1/s_mid), bias * fl(1/s_mid), s_mid, s_res, rh = fl(1/s_next), rl = fl(1/s_next - rh)
 * (i)  the first quotient becomes ONE fma on the accumulator with the existing margin test (|t - rint t| < 0.5 - 1e-4, else the exact
 *      division): valid when colscale is a power of two and |bias / s_mid| <= 512, which bounds the error of the folded form by 7e-5;
 * (ii) the second quotient becomes xs * rh + xs * rl (a 48-bit reciprocal) WITHOUT a test: its numerator takes at most 65 536 values per
 *      channel (256 residual codes x 256 q3 codes), and p2v_resid_prefold evaluates every one of them on the device against the IEEE
 *      division - the table is usable only if all N x 65 536 results agree.
 * *usable = 0 (table not to be used; the generic RESID epilogue gives the same codes) when (i) or (ii) does not hold.
 * Synchronises `stream`.  tab: dev, p2v_resid_prefold_bytes(N) bytes, owned by the caller, must outlive the launches that use it. */
size_t p2v_resid_prefold_bytes(int N);
int p2v_resid_prefold(const p2v_linear* lin, const p2v_epilogue* epi, int N, float* tab, size_t tab_bytes, int* usable, void* stream);

typedef struct p2v_plan p2v_plan;

/* ---- whole-model entry points: VisionTransformer.forward in quant state (vit_fquant.py:700-799) ---- */
int p2v_plan_create(const p2v_model_desc* desc, p2v_plan** out);
void p2v_plan_destroy(p2v_plan* plan);

/* layer index follows the reference's bit_config order (vit_fquant.py:710-711,748,789):
 * 0 = patch embed, 1+4*i+{0,1,2,3} = block i {qkv, proj, fc1, fc2}, 4*depth+1 = head.  bits in {4,8}. */
int p2v_plan_set_linear(p2v_plan* plan, int layer, int bits, const p2v_linear* lin);

/* inv_s_input = 1 / qact_input scale; inv_s_input == 0 selects VisionTransformer(input_quant=False) (the reference's vit_large
 * factory, vit_fquant.py:925): no input QAct, the fp32 image feeds the fake-quantised convolution directly - fp64 accumulation of the
 * exact products, one rounding to fp32; the layer-0 weights must then be unpacked codes (packed4 = 0) with colscale = s_w.
 * embed epilogue constants (one set per patch-embed bit width index 0:4-bit 1:8-bit is not
 * needed: activation scales do not depend on the weight bits); cls row codes [D] after qact1. */
int p2v_plan_set_embed(p2v_plan* plan, float inv_s_input, const p2v_epilogue* embed_epi,
                       const int8_t* cls_row_codes /* dev [D] */);

/* per block, per bit-pool index (0: 4-bit, 1: 8-bit) of qkv (for ln1/attn side) and of fc1 (for ln2 side):
 * the reference keeps best_scale/best_act_scale per bit (vit_fquant.py:282-292, layers_quant.py:305-313). */
typedef struct p2v_block {
  p2v_ln ln1[2];            /* indexed by bit-pool index of qkv */
  float inv_s_qkv[2];       /* 1/attn.qact1 scale (same for both, kept per index for symmetry) */
  p2v_attn attn;
  p2v_epilogue proj_epi;    /* RESID */
  p2v_ln ln2[2][2];         /* [bit index of qkv (attn.channel_scale quirk, vit_fquant.py:464)][bit index of fc1] */
  float inv_s_fc1;          /* 1/mlp.qact1 scale */
  p2v_gelu_tab gelu_fc1;    /* threshold table of fc1 -> GELU -> qact1 for inv_s_fc1 (table may be NULL) */
  p2v_epilogue fc2_epi;     /* RESID */
} p2v_block;
/* Stores the block's constants (the arrays stay the caller's and must outlive the plan) and - since round 3 - reads the LayerNorm arrays
 * back once to fold gamma / out_scale and beta / out_scale and to run the fast-chain tests on the host, so that the kernels of p2v_forward
 * do not repeat that per workgroup.  The call therefore synchronises the device (hipDeviceSynchronize) and must come after the arrays have
 * been written; changing an array afterwards requires setting the block again.  The folded copies are owned by the plan and live on the
 * device that owns blk->ln1[0].gamma (found with hipPointerGetAttributes - the process's current device does not matter; all blocks of a
 * plan must live on one device).  If the fold cannot be done (a HIP error, arrays on another device than earlier blocks) the call still
 * returns P2V_OK - the kernels then fold per workgroup, with identical results - and leaves the reason in p2v_last_error();
 * p2v_plan_block_prefolded() tells which of the two a block got. */
int p2v_plan_set_block(p2v_plan* plan, int block, const p2v_block* blk);
/* (Round 4: p2v_plan_set_block and p2v_plan_set_linear also build the plan's RESID tables - p2v_resid_prefold for proj / fc2 of a block once
 * both the block and the layer's weights are set; a later change of either rebuilds them.  blk->proj_epi.resid_tab / fc2_epi.resid_tab are
 * ignored.) */
/* 1: the LayerNorm constants of the block were folded when it was set; 0: its kernels fold per workgroup (p2v_last_error() of the
 * p2v_plan_set_block call says why); negative: error. */
int p2v_plan_block_prefolded(const p2v_plan* plan, int block);
/* which RESID tables of the block are in use: bit 0 proj / 4-bit weights, bit 1 proj / 8-bit, bit 2 fc2 / 4-bit, bit 3 fc2 / 8-bit
 * (a clear bit of a layer whose weights are set: p2v_resid_prefold found the pre-folded form not provable, the generic epilogue runs). */
int p2v_plan_resid_prefolded(const p2v_plan* plan, int block);

int p2v_plan_set_head(p2v_plan* plan, const p2v_ln* final_ln, float inv_s_out, float s_out);

size_t p2v_workspace_bytes(const p2v_plan* plan, int batch);

/* images: dev fp32 [batch][in_chans][img][img];  bit_config: HOST int8 [n_cfg], n_cfg = 4*depth+2;
 * logits: dev fp32 [batch][num_classes].  stop_after < 0 runs everything; otherwise execution stops after
 * that many kernel launches (parity tests read the workspace buffers, see p2v_workspace_view). */
int p2v_forward(p2v_plan* plan, const float* images, int batch, const int8_t* bit_config, int n_cfg,
                float* logits, void* workspace, size_t workspace_bytes, int stop_after, void* stream);

/* p2v_forward with the activation taps of the analysis scripts (cka_utility.py:26-113 reads blocks[i].attn.qkv_output and
 * blocks[i].mlp.fc1_output): qkv_out / fc1_out are HOST arrays of `depth` device pointers (entries or the arrays themselves may be
 * NULL); entry i receives fp32 [batch*tokens][3*embed_dim] / [batch*tokens][mlp_hidden] of block i. */
int p2v_forward_taps(p2v_plan* plan, const float* images, int batch, const int8_t* bit_config, int n_cfg,
                     float* logits, void* workspace, size_t workspace_bytes, float* const* qkv_out, float* const* fc1_out,
                     void* stream);

/* Kernel kinds reported by p2v_forward_profile. */
enum {
  P2V_K_PATCHIFY = 0, P2V_K_GEMM_EMBED = 1, P2V_K_FILL_CLS = 2, P2V_K_LAYERNORM = 3, P2V_K_GEMM_QKV = 4,
  P2V_K_ATTENTION = 5, P2V_K_GEMM_PROJ = 6, P2V_K_GEMM_FC1 = 7, P2V_K_GEMM_FC2 = 8, P2V_K_GEMM_HEAD = 9,
  P2V_K_LN_GEMM_QKV = 10, P2V_K_LN_GEMM_FC1 = 11,  /* LayerNorm fused into the GEMM that consumes it (p2v_ln_gemm_i8) */
  P2V_K_EVENT_GAP = 12     /* not a launch: the last interval of a profile pass, two events with NOTHING between them - what an event pair
                              itself adds to every interval on this stream (subtract it to compare with rocprofv3's kernel durations) */
};

/* Same as p2v_forward, with a hipEvent recorded on `stream` between consecutive launches; synchronises on the
 * last event (measurement only -- never call inside a timed region).  Returns the number of launches (>= 0) or
 * an error; ms_out[i] / kind_out[i] (HOST arrays, max_launches entries) receive the time from launch i to launch
 * i+1 on the stream and the P2V_K_* kind of launch i. */
int p2v_forward_profile(p2v_plan* plan, const float* images, int batch, const int8_t* bit_config, int n_cfg,
                        float* logits, void* workspace, size_t workspace_bytes, void* stream, float* ms_out,
                        int32_t* kind_out, int max_launches);

/* The same measurement for forwards that run CONCURRENTLY on several streams (the batch slices of the default step): _begin enqueues
 * the forward with its events on `stream` and returns at once with a token; _end synchronises on that forward's last event, fills
 * ms_out / kind_out like p2v_forward_profile, frees the token and returns the number of launches.  Begin every slice first, then end
 * them: ms_out[i] is then the duration of launch i WHILE the other streams' kernels run.  Every token must be ended; _end with
 * ms_out == NULL only waits for the forward and frees the token (error paths).  The events of a pass are created before its first
 * launch is enqueued (an event created between two launches would pace the host and the interval would measure the host), and the
 * returned count never exceeds max_launches. */
int p2v_forward_profile_begin(p2v_plan* plan, const float* images, int batch, const int8_t* bit_config, int n_cfg,
                              float* logits, void* workspace, size_t workspace_bytes, void* stream, void** token);
int p2v_forward_profile_end(void* token, float* ms_out, int32_t* kind_out, int max_launches);

/* byte offsets of the named activation buffers inside the workspace for `batch` ("patches", "x", "ln",
 * "qkv", "att", "hid", "cls"); returns <0 for unknown names. */
long long p2v_workspace_view(const p2v_plan* plan, int batch, const char* name);

/* ---- per-operator entry points (unit parity, and the module-level surface) --------------------------- */

/* QAct(qact_input) + im2col for the k=stride=patch QConv2d (vit_fquant.py:705-706, layers.py:55-88):
 * out[b*gh*gw + py*gw + px][c*P*P + i*P + j] = clamp(rne(img[b][c][py*P+i][px*P+j] * inv_s), -128, 127);
 * columns [C*P*P, k_pad) are zeroed. */
int p2v_quantize_patchify(const float* img, int batch, int chans, int height, int width, int patch,
                          float inv_s, int8_t* out, int k_pad, void* stream);

/* out = epilogue(A[M][K] . W[N][K]^T): int8 MFMA (v_mfma_i32_32x32x32_i8), fp32 epilogue in the
 * reference's operation order.  lda/ldo in elements.  `out` is int8 [M][ldo] except HEAD (fp32 [M][ldo]);
 * when out_codes != NULL the HEAD epilogue also writes the int8 logit codes there ([M][ldo]).
 * K is a multiple of 64 and MAY exceed lda (a width that is not a multiple of 64: the weight columns [width, K) are zero): the
 * contraction then walks into the next row, so (M-1)*lda + K bytes of A must be readable.  ldo >= N.  P2V_E_SHAPE otherwise. */
int p2v_gemm_i8(int epilogue_kind, const int8_t* A, int lda, int M, int K, int N, const p2v_linear* lin,
                const p2v_epilogue* epi, void* out, int ldo, int8_t* out_codes, void* stream);

/* rows x C int8 -> rows x C int8; row r of the input starts at x + r*row_stride (lets the final norm
 * touch only the cls rows, vit_fquant.py:766-767).  C: a multiple of 4 up to 2048; out_stride >= C. */
int p2v_int_layernorm(const int8_t* x, long long row_stride, int rows, int C, const p2v_ln* ln,
                      int8_t* out, long long out_stride, void* stream);

/* p2v_int_layernorm followed by p2v_gemm_i8 (REQUANT or GELU) in ONE launch: norm1 -> qkv -> qact1 and norm2 -> fc1 -> GELU -> qact1
 * (vit_fquant.py:431-434,284-293,307; layers_quant.py:305-316,331-333).  The LayerNorm output codes stay in LDS; ln_out (optional,
 * dev int8 [M][C]) also receives them.  x: dev int8, row r at x + r*row_stride, C channels; lin: weights [n_pad][round_up(C,64)].
 * Reads lin->w_frag (fragment-order weights), not lin->w_codes.  Bit-identical to the two separate calls.  P2V_E_UNSUPPORTED when
 * the shape is not instantiated (C > 384, or the layer's constants do not fit two workgroups per CU) or w_frag is NULL: run the
 * two calls instead. */
int p2v_ln_gemm_i8(int epilogue_kind, const int8_t* x, long long row_stride, int M, int C, const p2v_ln* ln, int N,
                   const p2v_linear* lin, const p2v_epilogue* epi, int8_t* out, int ldo, int8_t* ln_out, void* stream);
/* 1 when p2v_ln_gemm_i8 has an instantiation for this shape (epilogue REQUANT or GELU, C channels, N outputs, cells of the GELU table or
 * 0) in this process (the A/B switch P2V_LN_GEMM=0 turns every shape off), else 0: callers that record launch sequences ask first. */
int p2v_ln_gemm_fusable(int epilogue_kind, int C, int N, int gelu_table_cells);

/* fused attention core on the int8 qkv tensor [batch*tokens][3*heads*head_dim] (layout of
 * qkv.reshape(B,N,3,H,hd), vit_fquant.py:309-315); out int8 [batch*tokens][heads*head_dim].
 * probs_k (optional, dev int8 [batch][heads][tokens][tokens]) receives the log2 softmax exponents
 * (16 = zero) for parity tests. */
int p2v_lis_attention(const int8_t* qkv, int batch, int tokens, int heads, int head_dim, const p2v_attn* at,
                      int8_t* out, int8_t* probs_k, void* stream);

/* Swin window attention core (WindowAttention.forward between qact1 and qact3, swin_quant.py:186-217; window partition /
 * cyclic shift / reverse of SwinTransformerBlock.forward, swin_quant.py:366-391, folded into the addressing):
 *   (q*scale) @ k^T -> qact_attn1 -> + qact_table(relative_position_bias_table)[index] -> qact2 -> (+ -100 mask) ->
 *   QIntSoftmax(log-int, uint4, sf = qact2 scale) -> @ v -> qact3.
 * q*scale rounds each element once in fp32 and the dot product over head_dim is exact (fp64), then rounded once. */
typedef struct p2v_winattn {
  float s_q1;        /* qact1 scale (pot)                                        */
  float qk_scale;    /* head_dim^-0.5 (swin_quant.py:80)                         */
  float s_attn;      /* qact_attn1 scale (pot)                                   */
  float s_table;     /* qact_table scale (pot)                                   */
  float s_q2;        /* qact2 scale (pot) = the softmax scaling factor           */
  float s_q3;        /* qact3 scale (pot)                                        */
  int32_t x0_int, b_int, c_int; /* I-BERT constants for sf = s_q2 (layers.py:334-351) */
  const int8_t* table_codes;    /* dev [(2*ws-1)^2][heads] codes of qact_table(relative_position_bias_table) */
  const int32_t* win_index;     /* dev [n_windows][ws*ws]: row (within the image) of token p of window w after shift+partition */
  const int8_t* region;         /* dev [n_windows][ws*ws] shifted-window region ids, or NULL (no mask): pairs from different
                                 * regions get -100 (swin_quant.py:325-349)                                                   */
  int32_t ws, n_windows;
  int32_t qkv_stride;           /* bytes between qkv rows (0 = dense: 3*heads*head_dim)                  */
  int32_t out_stride;           /* bytes between out rows (0 = dense: heads*head_dim); lets the next GEMM read K padded to 64 */
} p2v_winattn;

/* qkv int8 [batch][tokens_per_image][3*heads*head_dim] (qact1 codes, natural token order); out int8
 * [batch][tokens_per_image][heads*head_dim] (qact3 codes, natural order).  head_dim must be 32.  probs_k optional:
 * dev int8 [batch][n_windows][heads][N][N] log2 exponents (16 = zero). */
int p2v_window_attention(const int8_t* qkv, int batch, int tokens_per_image, int heads, int head_dim,
                         const p2v_winattn* wa, int8_t* out, int8_t* probs_k, void* stream);

/* PatchMerging.forward gather (swin_quant.py:446-459): x int8 [batch][H*W][C] -> out int8 [batch][(H/2)*(W/2)][4*C] in the
 * reference's channel order x0 (even row, even col), x1 (odd row, even col), x2 (even row, odd col), x3 (odd, odd). */
int p2v_patch_merge_gather(const int8_t* x, int batch, int H, int W, int C, int8_t* out, void* stream);

/* SwinTransformer.forward_features tail (swin_quant.py:806-808): AdaptiveAvgPool1d over the tokens of the fake-quantised
 * qact2 output (codes * s_in, power-of-two s_in: the sum is exact), then qact3: out[b][c] = clamp(round((sum*s_in / tokens) * inv_s_out)). */
int p2v_avgpool_quant(const int8_t* x, int batch, int tokens, int C, float s_in, float inv_s_out, int8_t* out, void* stream);

/* A recorded sequence of the per-operator entry points above, replayed by ONE call: the host-side cost of a forward with a
 * few hundred launches (Swin: ~200) drops to one FFI crossing, and the sequence can be replayed on any stream.  Every record is
 * plain data; pointers are device pointers borrowed for the call.  kind selects which members are read:
 *   P2V_OP_PATCHIFY   in (fp32 images), out; i0 batch, i1 chans, i2 height, i3 width, i4 patch, i5 k_pad; f0 inv_s
 *   P2V_OP_GEMM       in (A), out; epi; M, K, N, lda, ldo; lin; ep
 *   P2V_OP_LAYERNORM  in, out; M rows, N channels, lda / ldo row strides; ln
 *   P2V_OP_WINATTN    in (qkv), out; i0 batch, i1 tokens per image, i2 heads, i3 head_dim; wa
 *   P2V_OP_MERGE      in, out; i0 batch, i1 H, i2 W, i3 C
 *   P2V_OP_AVGPOOL    in, out; i0 batch, i1 tokens, i2 C; f0 s_in, f1 inv_s_out
 *   P2V_OP_LN_GEMM    in (residual-stream codes), out; epi (REQUANT / GELU); M rows, K = C channels of the LayerNorm, N, lda row stride
 *                     of `in`, ldo == N; ln; lin (w_frag required); ep      = p2v_ln_gemm_i8: LayerNorm fused into the GEMM that reads it */
enum { P2V_OP_PATCHIFY = 0, P2V_OP_GEMM = 1, P2V_OP_LAYERNORM = 2, P2V_OP_WINATTN = 3, P2V_OP_MERGE = 4, P2V_OP_AVGPOOL = 5, P2V_OP_LN_GEMM = 6 };
typedef struct p2v_op {
  int32_t kind, epi;
  const void* in;
  void* out;
  int32_t M, K, N, lda, ldo;
  int32_t i0, i1, i2, i3, i4, i5;
  float f0, f1;
  p2v_linear lin;
  p2v_epilogue ep;
  p2v_ln ln;
  p2v_winattn wa;
} p2v_op;

/* run ops[0..n_ops) in order on `stream`; stops at the first error (its status is returned, p2v_last_error names the op). */
int p2v_run_ops(const p2v_op* ops, int n_ops, void* stream);
/* same, with HIP events recorded on `stream` around every op: ms[i] = duration of op i (synchronises the stream). */
int p2v_run_ops_profile(const p2v_op* ops, int n_ops, void* stream, float* ms);

/* UniformQuantizer.forward on an fp32 tensor (uniform.py:50-127, base.py:42-45): fake-quant in place of
 * the eager round/clamp chain.  scale has `n_scale` entries (1 = layer-wise) applied along the channel
 * dimension: element i uses scale[(i / inner) % n_scale].  codes (optional) receives the int8 codes. */
int p2v_fake_quant_f32(const float* x, long long n, const float* scale, int n_scale, long long inner,
                       int lo, int hi, float* out, int8_t* codes, void* stream);

/* Exact GELU -> requant threshold table for inv_s = 2^e, 0 <= e <= 12 (mlp.qact1 scale, layers_quant.py:331-333).
 *   p2v_gelu_table_plan   fills t->k, t->off, t->cells (t->table untouched); P2V_E_UNSUPPORTED when inv_s is not such a power
 *                         of two or the table would exceed 4096 cells (callers then leave table == NULL).
 *   p2v_gelu_table_build  t->table = dev buffer of t->cells * 8 bytes, scratch = dev buffer of p2v_gelu_table_scratch_bytes;
 *                         sweeps EVERY finite fp32 with the fp64 erfc on `stream` and SYNCHRONISES it (one-off, plan building):
 *                         P2V_E_UNSUPPORTED if some cell would need two thresholds.
 *   p2v_gelu_table_check  counts (into *mismatches, dev, zeroed by the caller) the finite fp32 values whose table code
 *                         differs from clamp(rne(RN32(gelu(y)) * inv_s)); asynchronous.                                    */
int p2v_gelu_table_plan(float inv_s, p2v_gelu_tab* t);
size_t p2v_gelu_table_scratch_bytes(int cells);
int p2v_gelu_table_build(float inv_s, const p2v_gelu_tab* t, void* scratch, size_t scratch_bytes, void* stream);
int p2v_gelu_table_check(float inv_s, const p2v_gelu_tab* t, unsigned long long* mismatches, void* stream);

/* correctly-rounded fp32 GELU followed by PoT quantisation (checks the fast path of the GELU epilogue
 * against its own fp64 slow path; flags[0] counts slow-path lanes). */
int p2v_gelu_quant_f32(const float* y, long long n, float inv_s, int8_t* codes, unsigned long long* flags,
                       int force_slow, void* stream);

/* max |fast GELU - fp64 GELU| over `count` consecutive fp32 bit patterns starting at first_bits; *max_err
 * (dev, fp32 bits compared as unsigned) must be zeroed by the caller.  Bound check for the epilogue. */
int p2v_gelu_err_sweep(unsigned first_bits, unsigned count, float* max_err, void* stream);

/* Scheduling aid, no data: enqueues `kernels` (1..64) launches of `workgroups` (1..4096) workgroups of 256 threads (with `lds_bytes`, 0..65536,
 * of LDS each) that wait `usec` (1..1000) microseconds on the device's constant-rate counter.  Two streams whose hardware queues are served
 * concurrently finish two such trains in little more than the time of one (10 x 20 us, 64 workgroups: 0.26 ms); streams whose queues share
 * a dispatch pipe, or that share a queue, take 0.46 - 0.60 ms - and a sliced forward on such a pair runs BELOW the one-stream rate
 * (profiles/r04_stream_pool.txt).  The host side probes candidate side streams with it before the first sliced forward (engine.side_streams). */
int p2v_stream_probe(void* stream, int kernels, int usec, int workgroups, int lds_bytes);

const char* p2v_last_error(void);
int p2v_abi_version(void);

/* Scheduling / A-B switches of the process (also read once from the environment: P2V_LN_GEMM, P2V_LN_GEMM_V, P2V_LN_GENERIC,
 * P2V_LN_ROWS, P2V_ATTN_WAVES, P2V_GEMM_TILE, P2V_RESID_PRE, P2V_LN_PRE, P2V_ATTN_STREAM).  None of them changes a result - every variant is
 * bit-identical and is driven through this call by the parity tests (profiles/r04_alt_paths.txt: the whole GPU suite on the alternatives):
 *   "ln_gemm" 0/1 (fuse LayerNorm into qkv / fc1), "ln_gemm_version" 1/2/3 (round-2 4-wave / pipelined 4-wave (default) / 8-wave fused kernel),
 *   "ln_generic" 0/1 (generic LayerNorm chain), "ln_rows" 1..64, "attn_waves" 4..8,
 *   "gemm_tile" 0/128/256 (tile height of the layer GEMMs; 0 = 256 rows when the grid still fills the chip),
 *   "resid_pre" 0/1 (use p2v_epilogue.resid_tab), "ln_pre" 0/1 (use p2v_ln.pre), "attn_stream" 0/1 (streaming attention kernel everywhere). */
int p2v_set_tuning(const char* name, int value);

#ifdef __cplusplus
}
#endif
#endif /* P2VIT_H */
