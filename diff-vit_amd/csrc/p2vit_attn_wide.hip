// p2vit_attn_wide.hip -- the ViT attention kernel for head dimensions beyond the DeiT / ViT-B / L family's 64 (round 4): 48, 80 (ViT-H), 96, 128.
// K (int8) and V^T (bf16) of an image's head stay in LDS - 3 * head_dim bytes per key - so the token limit falls with the head dimension:
// 608 tokens up to head_dim 80, 544 at 96, 384 at 128 (p2v_plan_create refuses beyond).
#include "p2vit_attn_lis.h"

int p2v_launch_attention_wide(const AttnArgs& a, int head_dim, int nkb, hipStream_t st) {
  if (head_dim == 48) { P2V_ATTN_CASES(48, 19, false) }
  if (head_dim == 80) { P2V_ATTN_CASES(80, 19, false) }
  if (head_dim == 96) { P2V_ATTN_CASES(96, 17, false) }
  if (head_dim == 128) { P2V_ATTN_CASES(128, 12, false) }
  return -1;
}
