// Thread -> voxel assignment of the voxel-loss kernels (pw_loss.hip, pw_loss2.hip).  The logits come with element strides: the
// reference's (B,C,X,Y,Z) tensor is, on this path, a permuted view of the OccHead's channels-last (B,Z,Y,X,C) buffer.  The kernels
// walk the voxels in the MEMORY order of the logits (spatial axes sorted by stride, smallest fastest), so consecutive lanes read
// consecutive rows; the dense (B,X,Y,Z) uint8 target / mask (640 KB, cache-resident) are the ones read out of order.  (Walking
// in target order put consecutive lanes 2.9 MB apart in the logits: 170-310 us per pass over 46 MB.)
#pragma once

struct VoxWalk { int p0, p1, p2; };      // spatial axis (0 = X, 1 = Y, 2 = Z) that is fastest / middle / slowest in memory

static inline VoxWalk vox_walk_order(long long sx, long long sy, long long sz) {
  int ax[3] = {2, 1, 0};                 // ties keep the target's own order (Z fastest)
  const long long st[3] = {sx, sy, sz};
  for (int i = 0; i < 3; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (st[ax[j]] < st[ax[i]]) { const int t = ax[i]; ax[i] = ax[j]; ax[j] = t; }
  VoxWalk w = {ax[0], ax[1], ax[2]};
  return w;
}

// u-th voxel of the walk -> batch b, coordinates (x, y, z), v = its index in the dense (B,X,Y,Z) target
template <class A>
__device__ __forceinline__ long long vox_walk(const A& a, long long u, int& b, int& x, int& y, int& z, long long& v) {
  const int n0 = a.walk.p0 == 0 ? a.X : (a.walk.p0 == 1 ? a.Y : a.Z);
  const int n1 = a.walk.p1 == 0 ? a.X : (a.walk.p1 == 1 ? a.Y : a.Z);
  const int n2 = a.walk.p2 == 0 ? a.X : (a.walk.p2 == 1 ? a.Y : a.Z);
  const int i0 = (int)(u % n0); long long t = u / n0;
  const int i1 = (int)(t % n1); t /= n1;
  const int i2 = (int)(t % n2);
  b = (int)(t / n2);
  x = a.walk.p0 == 0 ? i0 : (a.walk.p1 == 0 ? i1 : i2);
  y = a.walk.p0 == 1 ? i0 : (a.walk.p1 == 1 ? i1 : i2);
  z = a.walk.p0 == 2 ? i0 : (a.walk.p1 == 2 ? i1 : i2);
  v = (((long long)b * a.X + x) * a.Y + y) * a.Z + z;
  return b * a.sb + x * a.sx + y * a.sy + z * a.sz;
}
