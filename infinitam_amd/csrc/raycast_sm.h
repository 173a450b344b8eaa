// raycast_sm.h -- castRay for the hash index as a per-lane state machine with ONE memory round trip
// per wave iteration.
//
// Why: the straightforward march (raycast_device.h, a direct restatement of castRay,
// DeviceAgnostic/ITMVisualisationEngine.h:92-158) does, per step, up to three DEPENDENT memory round
// trips (occupancy bit/hash entry -> nearest voxel -> neighbour blocks -> eight trilinear voxels) and a
// wave pays the longest chain of any of its 64 lanes on every iteration.  rocprofv3 PMC on MI355X:
// 63 % of the wave cycles are s_waitcnt, ~3 700 cycles per wave iteration, kernel time = (steps of the
// longest ray) x that.  Here every lane is in exactly one state per iteration and ALL loads of all
// states are issued before the single wait:
//
//   ST_LOOKUP   : probe the table for the blocks in `need` (<= 8 slots of the 2x2x2 block
//                 neighbourhood of the anchor block = block of floor(p)); occupancy bit and 16-byte
//                 entry are fetched together; excess chains continue in the next iteration
//   ST_READ     : nearest-voxel read; if the whole 2x2x2 voxel neighbourhood lies in known blocks its
//                 8 values are fetched in the same iteration (they serve the trilinear re-read)
//   ST_TRI_READ : the 8 voxel loads of a trilinear read whose blocks had to be looked up first
//
// A lane needs more iterations per ray step than before (a new block costs one extra iteration),
// but an iteration is one round trip instead of up to three and lanes no longer wait for each
// other's chains.  Positions visited, values read and all float arithmetic are exactly those of the
// reference (bit-exact vs the CPU oracle); only the order in which independent reads are issued
// changes.
#pragma once

#include "raycast_device.h"

namespace itm {

template <class VX>
__device__ inline float4 cast_ray_sm(int x, int y, const VolumeView& vol, const RayParams& p, float2 mm) {
  enum : int { ST_LOOKUP = 0, ST_READ = 1, ST_TRI_READ = 2, ST_DONE = 3 };
  const float stepScale = p.mu * p.oneOverVoxel;
  // ---- ray set-up: identical arithmetic to cast_ray ------------------------------------------
  float pcz = mm.x;
  float pcx = pcz * (((float)x - p.cx) * p.ifx);
  float pcy = pcz * (((float)y - p.cy) * p.ify);
  float acc = 0.0f; acc += pcx * pcx; acc += pcy * pcy; acc += pcz * pcz;
  float total = sqrtf(acc) * p.oneOverVoxel;
  Vec3 t = transform_point(p.invM, pcx, pcy, pcz);
  const float sx = t.x * p.oneOverVoxel, sy = t.y * p.oneOverVoxel, sz = t.z * p.oneOverVoxel;
  pcz = mm.y;
  pcx = pcz * (((float)x - p.cx) * p.ifx);
  pcy = pcz * (((float)y - p.cy) * p.ify);
  acc = 0.0f; acc += pcx * pcx; acc += pcy * pcy; acc += pcz * pcz;
  const float totalMax = sqrtf(acc) * p.oneOverVoxel;
  t = transform_point(p.invM, pcx, pcy, pcz);
  float dx = t.x * p.oneOverVoxel - sx, dy = t.y * p.oneOverVoxel - sy, dz = t.z * p.oneOverVoxel - sz;
  const float dn = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
  dx *= dn; dy *= dn; dz *= dn;
  float px = sx, py = sy, pz = sz;
  float sdf = 1.0f;
  const float dflt = VX::kShort ? 32767.0f : 1.0f;

  // ---- per-lane machine state ------------------------------------------------------------------
  int state = ST_DONE;
  bool refine = false;         // false: marching; true: the trilinear read of the post-hit refinement
  bool hit = false;
  int ax = 0x7fffffff, ay = 0x7fffffff, az = 0x7fffffff;   // anchor block = block of floor(p)
  int base[8];                 // voxel base of slot s = anchor + (s&1, s>>1&1, s>>2), -1 = not allocated
  int probe[8];                // next table index to probe for slot s (chain continuation), -1 = start at the bucket
  int known = 0;               // slots whose base[] is final
  int need = 0;                // slots ST_LOOKUP has to resolve
  int after = ST_READ;         // state entered when `need` is resolved
  int lastx = 0x7fffffff, lasty = 0x7fffffff, lastz = 0x7fffffff, lastBase = -1;  // last block found (IndexCache)
  // geometry of the current position
  int lx = 0, ly = 0, lz = 0;  // floor(p) inside the anchor block
  int cross = 0;               // bit k: +1 along axis k leaves the anchor block
  int sn = 0, linN = 0;        // slot and in-block offset of the nearest voxel
  float fcx = 0, fcy = 0, fcz = 0;  // fractional position
#pragma unroll
  for (int s = 0; s < 8; ++s) { base[s] = -1; probe[s] = -1; }

  // (re)derives anchor, nearest voxel and crossing flags for the current p; keeps resolved slots
  // while the anchor block does not change
  auto locate = [&]() {
    const float flx = floorf(px), fly = floorf(py), flz = floorf(pz);
    fcx = px - flx; fcy = py - fly; fcz = pz - flz;
    const int ix = (int)flx, iy = (int)fly, iz = (int)flz;
    const int bx = ix >> 3, by = iy >> 3, bz = iz >> 3;   // == ((v<0)? v-7 : v)/8
    lx = ix & 7; ly = iy & 7; lz = iz & 7;
    cross = (lx == 7 ? 1 : 0) | (ly == 7 ? 2 : 0) | (lz == 7 ? 4 : 0);
    if (bx != ax || by != ay || bz != az) {
      ax = bx; ay = by; az = bz; known = 0;
#pragma unroll
      for (int s = 0; s < 8; ++s) probe[s] = -1;
    }
    const int nx = (int)round_ref(px), ny = (int)round_ref(py), nz = (int)round_ref(pz);
    const int dnb = (nx - ix) | ((ny - iy) << 1) | ((nz - iz) << 2);   // ROUND(p) - floor(p) in {0,1}^3
    sn = dnb & cross;
    linN = (nx & 7) + (ny & 7) * 8 + (nz & 7) * 64;
  };
  // slots touched by the 2x2x2 voxel neighbourhood
  auto corner_slots = [&]() { int m = 1; if (cross & 1) m |= m << 1; if (cross & 2) m |= m << 2; if (cross & 4) m |= m << 4; return m; };
  // after p changed: choose the next state
  auto enter_position = [&]() {
    locate();
    const int want = refine ? corner_slots() : (1 << sn);
    need = want & ~known;
    after = refine ? ST_TRI_READ : ST_READ;
    state = need ? ST_LOOKUP : after;
  };
  auto advance = [&](float step) {
    px += step * dx; py += step * dy; pz += step * dz;
    total += step;
    if (total < totalMax) enter_position(); else state = ST_DONE;
  };
  auto blend = [&](const float* v) {
    float r1 = (1.0f - fcx) * v[0] + fcx * v[1];
    r1 = (1.0f - fcy) * r1 + fcy * ((1.0f - fcx) * v[2] + fcx * v[3]);
    float r2 = (1.0f - fcx) * v[4] + fcx * v[5];
    r2 = (1.0f - fcy) * r2 + fcy * ((1.0f - fcx) * v[6] + fcx * v[7]);
    return VX::to_float((1.0f - fcz) * r1 + fcz * r2);
  };
  // consumes an interpolated value: refinement end, surface crossing or next step
  auto after_trilinear = [&](float v) {
    sdf = v;
    if (refine) {                       // second refinement move, then done
      const float step = sdf * stepScale;
      px += step * dx; py += step * dy; pz += step * dz;
      hit = true; state = ST_DONE;
    } else if (sdf <= 0.0f) {           // surface crossed: first refinement move + trilinear re-read
      const float step = sdf * stepScale;
      px += step * dx; py += step * dy; pz += step * dz;
      refine = true;
      enter_position();
    } else {
      const float s = sdf * stepScale;
      advance((s < 1.0f) ? 1.0f : s);
    }
  };

  if (total < totalMax) enter_position();

  while (__any(state != ST_DONE)) {
    // ================= issue phase: every load of this iteration ===============================
    uint32_t bits[8];
    uint4 ent[8];
    bool asked[8];
    const bool inLookup = state == ST_LOOKUP;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      asked[s] = inLookup && ((need >> s) & 1);
      if (asked[s]) {
        const int bx = ax + (s & 1), by = ay + ((s >> 1) & 1), bz = az + (s >> 2);
        if (probe[s] < 0 && bx == lastx && by == lasty && bz == lastz) {   // IndexCache hit: no memory
          base[s] = lastBase; known |= 1 << s; need &= ~(1 << s); asked[s] = false;
        } else {
          const int first = probe[s] < 0;
          const int idx = first ? hash_index(bx, by, bz, vol.mask) : probe[s];
          bits[s] = first ? vol.headBits[idx >> 5] >> (idx & 31) : 1u;
          ent[s] = vol.hash[idx];
        }
      }
    }
    const bool inRead = state == ST_READ;
    const bool readHasBlock = inRead && base[sn] >= 0;
    const int cs = corner_slots();
    // the 8 corners can ride along with the nearest read when all their blocks are already known
    const bool cornersNow = (state == ST_TRI_READ) || (readHasBlock && ((cs & ~known) == 0));
    float rawN = dflt;
    if (readHasBlock) rawN = VX::load_raw_sdf(vol.vba, (size_t)(base[sn] + linN));
    float cv[8];
    if (cornersNow) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int b = base[c & cross];
        const int off = ((lx + (c & 1)) & 7) + ((ly + ((c >> 1) & 1)) & 7) * 8 + ((lz + (c >> 2)) & 7) * 64;
        cv[c] = (b >= 0) ? VX::load_raw_sdf(vol.vba, (size_t)(b + off)) : dflt;
      }
    }

    // ================= resolve phase (first use of the loaded values = the single wait) =========
    if (inLookup) {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if (asked[s]) {
          const int bx = ax + (s & 1), by = ay + ((s >> 1) & 1), bz = az + (s >> 2);
          const HashEntry e = unpack_entry(ent[s]);
          bool done = true; int b = -1;
          if (bits[s] & 1u) {
            if (e.px == bx && e.py == by && e.pz == bz && e.ptr >= 0) {
              b = e.ptr * kBlockVoxels;
              lastx = bx; lasty = by; lastz = bz; lastBase = b;
            } else if (e.offset >= 1) {
              probe[s] = vol.bucketNum + e.offset - 1;   // follow the excess chain next iteration
              done = false;
            }
          }
          if (done) { base[s] = b; known |= 1 << s; need &= ~(1 << s); probe[s] = -1; }
        }
      }
      if (need == 0) state = after;
    } else if (inRead) {
      if (!readHasBlock) {
        sdf = VX::to_float(dflt);                 // block missing: default voxel, step one block
        advance((float)kBlockSide);
      } else {
        sdf = VX::to_float(rawN);
        if ((sdf <= 0.1f) && (sdf >= -0.5f)) {
          if (cornersNow) after_trilinear(blend(cv));
          else { need = cs & ~known; after = ST_TRI_READ; state = need ? ST_LOOKUP : ST_TRI_READ; }
        } else if (sdf <= 0.0f) {                 // below the band: surface crossed on the nearest value
          const float step = sdf * stepScale;
          px += step * dx; py += step * dy; pz += step * dz;
          refine = true;
          enter_position();
        } else {
          const float s = sdf * stepScale;
          advance((s < 1.0f) ? 1.0f : s);
        }
      }
    } else if (state == ST_TRI_READ) {
      after_trilinear(blend(cv));
    }
  }
  return make_float4(px, py, pz, hit ? 1.0f : 0.0f);
}

#ifndef ITM_RAY_STATE_MACHINE
#define ITM_RAY_STATE_MACHINE 0   // measured 4x SLOWER (299 vs 70 us): the kernel is instruction-issue bound at ~17 % lane utilisation, see DESIGN.md
#endif

// entry point used by the kernels: state machine for the hash index, direct march for the dense array
template <class VX, bool DENSE>
__device__ inline float4 cast_ray_any(int x, int y, const VolumeView& vol, const RayParams& p, float2 mm) {
#if ITM_RAY_STATE_MACHINE
  if constexpr (!DENSE) return cast_ray_sm<VX>(x, y, vol, p, mm);
  else return cast_ray<VX, DENSE>(x, y, vol, p, mm);
#elif ITM_RAY_WHILE_WHILE
  return cast_ray_ww<VX, DENSE>(x, y, vol, p, mm);
#else
  return cast_ray<VX, DENSE>(x, y, vol, p, mm);
#endif
}

}  // namespace itm
