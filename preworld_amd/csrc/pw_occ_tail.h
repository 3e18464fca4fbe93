// Tail of the fused OccHead kernels (pw_occ_head.hip, pw_occ_head_h2.hip): what follows the 3x3x3 conv per voxel
// (mmdet3d/models/heads/occupancy_head.py:92-99,124-161): 1x1x1 16->8 + BN + ReLU, 1x1x1 8->18, argmax -> uint8.
#ifndef PW_OCC_TAIL_H_
#define PW_OCC_TAIL_H_
#include <stdint.h>

struct OccTail {
  const float* w1;      // [8][16]  occ_pred_conv.0.weight
  const float* s1;      // [8]      folded BN scale
  const float* b1;      // [8]      folded BN bias
  const float* w2;      // [18][8]  occ_pred_conv.3.weight
  uint8_t* occ;         // [B*D*H*W] argmax class
  float* logits;        // [B*D*H*W][18] or null
  uint8_t* geo;         // [B*D*H*W] geo_occ or null
  int empty_idx;
  int n_mid, n_hid, n_cls;
  // k_occ_head_h2 only (pw_occ_head_h2_strided): byte strides of occ / geo along (b, d, h, w) and the span of the buffers they live in;
  // contiguous (B, D, H, W): (D H W, H W, W, 1) and B D H W
  int sb, sd, sh, sw;
  unsigned span;
};

#endif  // PW_OCC_TAIL_H_
