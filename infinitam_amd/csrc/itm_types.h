// itm_types.h -- device/host POD layouts of the TSDF path, byte-identical to the reference
// (Utils/ITMLibDefines.h:71-82 ITMHashEntry, :100-199 ITMVoxel_*), plus the voxel codec traits
// that replace the reference's TVoxel template parameter.  Padding bytes are always written as 0.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace itm {

constexpr int kBlockSide = 8;     // SDF_BLOCK_SIZE
constexpr int kBlockVoxels = 512; // SDF_BLOCK_SIZE3

// 16-byte hash entry: {short pos[3]; (pad); int offset; int ptr}.  Loaded/stored as one uint4.
struct __attribute__((aligned(16))) HashEntry {
  int16_t px, py, pz, pad;
  int32_t offset;  // 1-based index into the excess region, 0 = end of chain
  int32_t ptr;     // >=0 voxel block, -1 swapped out, <-1 free
};
static_assert(sizeof(HashEntry) == 16, "hash entry layout");

__host__ __device__ inline HashEntry unpack_entry(const uint4& r) {
  HashEntry e;
  e.px = (int16_t)(r.x & 0xffffu);
  e.py = (int16_t)(r.x >> 16);
  e.pz = (int16_t)(r.y & 0xffffu);
  e.pad = 0;
  e.offset = (int32_t)r.z;
  e.ptr = (int32_t)r.w;
  return e;
}
__host__ __device__ inline uint4 pack_entry(int bx, int by, int bz, int offset, int ptr) {
  uint4 r;
  r.x = ((uint32_t)(uint16_t)(int16_t)bx) | (((uint32_t)(uint16_t)(int16_t)by) << 16);
  r.y = ((uint32_t)(uint16_t)(int16_t)bz);
  r.z = (uint32_t)offset;
  r.w = (uint32_t)ptr;
  return r;
}

// hashIndex (DeviceAgnostic/ITMRepresentationAccess.h:8-10): coordinates sign-extend to uint32.
__host__ __device__ inline int hash_index(int bx, int by, int bz, uint32_t mask) {
  return (int)((((uint32_t)bx * 73856093u) ^ ((uint32_t)by * 19349669u) ^ ((uint32_t)bz * 83492791u)) & mask);
}

// x / 32767.0f, correctly rounded, in three instructions instead of the ~12 of the IEEE division
// macro: q = RN(x*r), e = x - q*d (exact in one FMA), q' = RN(q + e*r) with r = RN(1/d).  This is
// Markstein's reciprocal-based division; for a divisor whose significand is not all ones (32767 =
// 0x7FFF is 15 one-bits in a 24-bit significand) q' equals the correctly rounded quotient for every
// x whose quotient is a normal number (|x| >= 2^-111), which covers every value on this path.
// tests/test_hip_parity.py::test_div_by_32767_is_ieee checks it against the division exhaustively
// over all 2^16 short values and a dense sample of floats.
#ifndef ITM_FAST_DIV32767
#define ITM_FAST_DIV32767 1
#endif
__device__ inline float div_by_32767(float x) {
#if ITM_FAST_DIV32767
  const float d = 32767.0f;
  const float r = 1.0f / 32767.0f;   // constant-folded, correctly rounded
  const float q = x * r;
  const float e = __builtin_fmaf(-q, d, x);
  return __builtin_fmaf(e, r, q);
#else
  return x / 32767.0f;
#endif
}

// ---- correctly rounded divisions without the IEEE division macro ---------------------------------
// hipcc lowers a/b (f32, IEEE) to v_div_scale x2, v_rcp, five FMAs, v_div_fmas, v_div_fixup.  The scale /
// fixup steps only act on denormal, huge or special operands; for operands in a normal range the result is
// exactly the FMA chain below, so replaying that chain gives bit-identical quotients with fewer
// instructions, and a refined reciprocal can be shared by several divisions by the same denominator.
//   r  = rcp(b);  r = r + r*(1 - b*r)                      (refined reciprocal, 3 instructions)
//   q  = a*r;  q = q + r*(a - b*q);  q = q + r*(a - b*q)   (5 instructions per quotient)
// Callers guarantee b is a normal number well inside the exponent range (guards at the call sites);
// tests/test_hip_parity.py::test_fast_divisions_are_ieee compares all helpers with the true division.
#ifndef ITM_FAST_DIVISIONS
#define ITM_FAST_DIVISIONS 1
#endif
__device__ inline float refined_rcp(float b) {
  const float r = __builtin_amdgcn_rcpf(b);
  const float e = __builtin_fmaf(-b, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}
__device__ inline float div_by_rcp(float a, float b, float r) {
  float q = a * r;
  float e = __builtin_fmaf(-b, q, a);
  q = __builtin_fmaf(e, r, q);
  e = __builtin_fmaf(-b, q, a);
  return __builtin_fmaf(e, r, q);
}
// a / b when r == RN(1/b) exactly and the significand of b is not all ones (Markstein): 3 instructions
__device__ inline float div_markstein(float a, float b, float r) {
  const float q = a * r;
  const float e = __builtin_fmaf(-q, b, a);
  return __builtin_fmaf(e, r, q);
}

// ---- voxel codecs ---------------------------------------------------------------------------
// Each codec describes one ITMVoxel_* layout through a register image (`Reg`) that is moved with
// the widest aligned access the layout allows, and decoded/encoded field-wise.
struct VoxelS {  // ITMVoxel_s: {i16 sdf @0; u8 w_depth @2; pad @3}, 4 B
  static constexpr int kBytes = 4;
  static constexpr bool kColor = false;
  static constexpr bool kShort = true;
  using Reg = uint32_t;
  __device__ static Reg load(const void* base, size_t i) { return ((const uint32_t*)base)[i]; }
  __device__ static void store(void* base, size_t i, Reg r) { ((uint32_t*)base)[i] = r; }
  __device__ static float raw_sdf(Reg r) { return (float)(int16_t)(r & 0xffffu); }
  __device__ static int w_depth(Reg r) { return (int)((r >> 16) & 0xffu); }
  __device__ static float to_float(float raw) { return div_by_32767(raw); }
  __device__ static Reg with_depth(Reg, float f, int w) {
    int16_t s = (int16_t)(f * 32767.0f);
    return ((uint32_t)(uint16_t)s) | ((uint32_t)(w & 0xff) << 16);
  }
  __device__ static Reg init() { return 32767u; }
  // nearest/trilinear reads only need the sdf: a 2-byte load
  __device__ static float load_raw_sdf(const void* base, size_t i) { return (float)((const int16_t*)base)[i * 2]; }
};

struct VoxelF {  // ITMVoxel_f: {f32 sdf @0; u8 w_depth @4; pad}, 8 B
  static constexpr int kBytes = 8;
  static constexpr bool kColor = false;
  static constexpr bool kShort = false;
  using Reg = uint2;
  __device__ static Reg load(const void* base, size_t i) { return ((const uint2*)base)[i]; }
  __device__ static void store(void* base, size_t i, Reg r) { ((uint2*)base)[i] = r; }
  __device__ static float raw_sdf(Reg r) { return __uint_as_float(r.x); }
  __device__ static int w_depth(Reg r) { return (int)(r.y & 0xffu); }
  __device__ static float to_float(float raw) { return raw; }
  __device__ static Reg with_depth(Reg, float f, int w) { return make_uint2(__float_as_uint(f), (uint32_t)(w & 0xff)); }
  __device__ static Reg init() { return make_uint2(__float_as_uint(1.0f), 0u); }
  __device__ static float load_raw_sdf(const void* base, size_t i) { return ((const float*)base)[i * 2]; }
};

struct VoxelSRgb {  // ITMVoxel_s_rgb: {i16 sdf @0; u8 w_depth @2; u8 clr[3] @3; u8 w_color @6; pad @7}, 8 B
  static constexpr int kBytes = 8;
  static constexpr bool kColor = true;
  static constexpr bool kShort = true;
  using Reg = uint2;
  __device__ static Reg load(const void* base, size_t i) { return ((const uint2*)base)[i]; }
  __device__ static void store(void* base, size_t i, Reg r) { ((uint2*)base)[i] = r; }
  __device__ static float raw_sdf(Reg r) { return (float)(int16_t)(r.x & 0xffffu); }
  __device__ static int w_depth(Reg r) { return (int)((r.x >> 16) & 0xffu); }
  __device__ static float to_float(float raw) { return div_by_32767(raw); }
  __device__ static Reg with_depth(Reg r, float f, int w) {
    int16_t s = (int16_t)(f * 32767.0f);
    r.x = (r.x & 0xff000000u) | ((uint32_t)(uint16_t)s) | ((uint32_t)(w & 0xff) << 16);
    return r;
  }
  __device__ static void get_color(Reg r, int c[3], int& wc) {
    c[0] = (int)(r.x >> 24); c[1] = (int)(r.y & 0xffu); c[2] = (int)((r.y >> 8) & 0xffu); wc = (int)((r.y >> 16) & 0xffu);
  }
  __device__ static Reg with_color(Reg r, const int c[3], int wc) {
    r.x = (r.x & 0x00ffffffu) | ((uint32_t)(c[0] & 0xff) << 24);
    r.y = (uint32_t)(c[1] & 0xff) | ((uint32_t)(c[2] & 0xff) << 8) | ((uint32_t)(wc & 0xff) << 16);
    return r;
  }
  __device__ static Reg init() { return make_uint2(32767u, 0u); }
  __device__ static float load_raw_sdf(const void* base, size_t i) { return (float)((const int16_t*)base)[i * 4]; }
};

struct VoxelFRgb {  // ITMVoxel_f_rgb: {f32 sdf @0; u8 w_depth @4; u8 clr[3] @5; u8 w_color @8; pad}, 12 B
  static constexpr int kBytes = 12;
  static constexpr bool kColor = true;
  static constexpr bool kShort = false;
  struct Reg { uint32_t a, b, c; };
  __device__ static Reg load(const void* base, size_t i) {
    const uint32_t* p = (const uint32_t*)base + i * 3;
    Reg r; r.a = p[0]; r.b = p[1]; r.c = p[2]; return r;
  }
  __device__ static void store(void* base, size_t i, Reg r) {
    uint32_t* p = (uint32_t*)base + i * 3;
    p[0] = r.a; p[1] = r.b; p[2] = r.c;
  }
  __device__ static float raw_sdf(Reg r) { return __uint_as_float(r.a); }
  __device__ static int w_depth(Reg r) { return (int)(r.b & 0xffu); }
  __device__ static float to_float(float raw) { return raw; }
  __device__ static Reg with_depth(Reg r, float f, int w) {
    r.a = __float_as_uint(f);
    r.b = (r.b & 0xffffff00u) | (uint32_t)(w & 0xff);
    return r;
  }
  __device__ static void get_color(Reg r, int c[3], int& wc) {
    c[0] = (int)((r.b >> 8) & 0xffu); c[1] = (int)((r.b >> 16) & 0xffu); c[2] = (int)(r.b >> 24); wc = (int)(r.c & 0xffu);
  }
  __device__ static Reg with_color(Reg r, const int c[3], int wc) {
    r.b = (r.b & 0xffu) | ((uint32_t)(c[0] & 0xff) << 8) | ((uint32_t)(c[1] & 0xff) << 16) | ((uint32_t)(c[2] & 0xff) << 24);
    r.c = (uint32_t)(wc & 0xff);
    return r;
  }
  __device__ static Reg init() { Reg r; r.a = __float_as_uint(1.0f); r.b = 0; r.c = 0; return r; }
  __device__ static float load_raw_sdf(const void* base, size_t i) { return ((const float*)base)[i * 3]; }
};

// ---- block directory ------------------------------------------------------------------------
// A dense mirror of the hash table over the block coordinates [-kDirHalf, kDirHalf)^3, spending HBM capacity (512 MB of
// 288 GB) to turn "is block b allocated, and where" into ONE load from a spatially coherent array instead of an occupancy
// bit + hash entry (+ chain) from spatially incoherent ones:
//   dirPtr[cell]   int32: voxel-block index (ITMHashEntry::ptr) of the block at that position, -1 if none.  Cells are
//                  stored brick-major: a brick = 4x4x4 blocks = 64 cells = 256 contiguous bytes, so the rays of a wave
//                  (neighbouring pixels) read one or two cache lines per step.
// The cube is NOT tied to the world origin: AccelOrigin (below) holds the block coordinate of its cell (0, 0, 0), placed around the
// camera when the first frame arrives and moved when the camera leaves (scene.hip, accel_place): the reference's table has no
// spatial limit (Objects/ITMVoxelBlockHash.h:22-100), and the frame an external pose source uses is not ours to choose.
// Blocks outside the covered cube are looked up through the hash table as before.  The array is written by the allocation
// sweep for every block it allocates (and rebuilt from the table after an upload), so it holds exactly the entries with
// ptr >= 0 -- the ones the reference's readVoxel finds (DeviceAgnostic/ITMRepresentationAccess.h:85-119).
// (A second level -- a 64-bit brick-occupancy word per 16^3-block super-brick kept in registers, so that rays cross empty
// bricks on arithmetic alone, with joint "runs" of the lanes of a wave and provably-safe multi-step advances -- was built
// and measured: 83-104 us against 64 us for the plain directory; the classification arithmetic per step costs more than
// the L1-resident load it saves and the extra loop structure de-synchronises the lanes.  Removed)
constexpr int kDirBits = 9;
constexpr int kDirSide = 1 << kDirBits;        // 512 blocks per axis: +-8.2 m at 4 mm voxels, +-4.1 m at 2 mm
constexpr int kDirHalf = kDirSide / 2;
constexpr size_t kDirCells = (size_t)kDirSide * kDirSide * kDirSide;

// Where the two acceleration cubes lie in the world: block coordinates of the directory's cell (0, 0, 0) and of the mirror's.
// Passed to kernels by value; changed only by the host between launches (scene.hip: accel_place re-fills the cubes around it).
struct AccelOrigin {
  int32_t dx, dy, dz;
  int32_t mx, my, mz;
  // the sdf mirror (below).  mMaxPages > 0: PAGED -- the page table, the pool's page counter and size; mMaxPages < 0: DENSE -- a whole
  // cube of 2^(-mMaxPages) blocks per side is stored (-8: 256^3 blocks, -7: 128^3; mirror_dense_bits), no table (mTable / mPages are
  // nullptr); 0: the scene has no mirror
  int32_t* mTable;
  int32_t* mPages;
  int32_t mMaxPages;
};

// cube-relative block coordinates (each in [0, kDirSide) when the block is covered)
__host__ __device__ inline bool dir_covers(uint32_t ux, uint32_t uy, uint32_t uz) { return ((ux | uy | uz) >> kDirBits) == 0u; }
// brick-major: 4 x 4 x 4 blocks are 256 contiguous bytes (the 2 x 2 x 2 block neighbourhood of a trilinear read mostly lies in one)
__host__ __device__ inline uint32_t dir_cell(uint32_t ux, uint32_t uy, uint32_t uz) {
  const uint32_t brick = ((uz >> 2) << (2 * (kDirBits - 2))) | ((uy >> 2) << (kDirBits - 2)) | (ux >> 2);
  return (brick << 6) | ((uz & 3u) << 4) | ((uy & 3u) << 2) | (ux & 3u);
}
// records an allocated block (device side; called by the allocation sweep and the rebuild kernel)
// (dirSlot: the same cells holding the TABLE SLOT of the block instead of its voxel-block index -- what the allocation request
// needs to mark a block that already exists as visible, one coherent 4-byte load instead of the 16-byte entry of a random bucket)
__device__ inline void directory_insert(int32_t* __restrict__ dirPtr, int32_t* __restrict__ dirSlot, const AccelOrigin& org, int bx, int by, int bz, int ptr, int slot) {
  const uint32_t ux = (uint32_t)(bx - org.dx), uy = (uint32_t)(by - org.dy), uz = (uint32_t)(bz - org.dz);
  if (dirPtr && dir_covers(ux, uy, uz)) {          // a scene may run without the directories (scene.hip)
    const uint32_t cell = dir_cell(ux, uy, uz);
    dirPtr[cell] = ptr;
    if (dirSlot) dirSlot[cell] = slot;
  }
}

// ---- sdf mirror -------------------------------------------------------------------------------
// A second copy of the sdf of every voxel of every allocated block, addressed by POSITION instead of through the block pointer, over a
// cube of 256^3 blocks placed in front of the camera (AccelOrigin: centred kMirrorShift blocks along the viewing direction of the frame
// that placed it, moved when the view leaves it).  Rounds 2-3 stored the cube densely -- 256^3 cells x 1 KB = 17 GB per scene whatever
// it held.  Round 4: the cube is PAGED.  A page is 16 x 16 x 16 blocks (4 MB of int16 sdf, 8 MB of float bits), its blocks
// x-fastest at a kilobyte each, a block's voxels in the block's own order (mirror_in_page).  A table of
// 16^3 entries (16 KB: every ray-cast workgroup keeps a copy in LDS) says for every page of the cube
//     >= 0   the page's index in the pool: value = pool[page << 21 | place in the page]
//     -1     no block was ever allocated in the page: every position in it reads "no block" WITHOUT a second load
//     -3     the pool had run dry when a block of the page was allocated: the page says nothing, readers use the block directory
//     (-2    a thread is taking a page from the pool right now; only ever seen inside the kernel that allocates)
// and pages are handed out by whoever allocates the first block in them (mirror_claim_page).  Memory is O(touched pages): the bench
// scene (sphere + wall, 60 k blocks allocated over the trajectory) maps a few dozen pages of a 768 MB pool.  What a ray step costs: the
// table entry (kept per lane while the ray stays inside the page -- a page is 128 voxels wide, a step at most 8 -- and otherwise read
// from LDS) and then ONE load whose address follows from the position, as before; in empty space the table alone answers, where the
// dense cube answered with a cold kilobyte of HBM per cell.
// BOTH forms exist at run time (AccelOrigin::mMaxPages): the dense cube is the faster one -- ray cast 38.3 us against 42.8-43.4 paged on
// BASELINE configs[1]: a page's address takes ~12 more vector instructions per step and the table entry is one more dependent load
// whenever a lane changes page (profiles/r4_raycast_notes.md) -- and is taken while the device has three times its 17 GB to spare;
// every further scene, and every scene created with ITM_MIRROR=paged in the environment, gets the paged form.  "Absent" inside a mapped page: -32768 cannot be a stored short sdf
// ((short)(f * 32767) with f in [-1, 1]); 0xFFFFFFFF is a NaN no arithmetic produces.  Written wherever voxels are written: at
// allocation (the initial value), by the integration, by the swapping engine, and again from the table after the cube has moved or
// the table was replaced.  Invariant: the only cells of mapped pages that are not "absent" are those of table entries with ptr >= 0 --
// so emptying the mirror is a pass over the table that writes "absent" into exactly those cells, after which EVERY page of the pool
// is clean again, the page table returns to -1 and the pool's counter to 0.
constexpr int kMirrorBits = 8;
constexpr int kMirrorSide = 1 << kMirrorBits;
constexpr int kMirrorHalf = kMirrorSide / 2;
constexpr int kMirrorShift = kMirrorSide / 4;      // the cube is centred kMirrorShift blocks in front of the camera that placed it
constexpr size_t kMirrorCells = (size_t)kMirrorSide * kMirrorSide * kMirrorSide;
// log2 of a page's side in blocks.  Measured (ray cast in frame, BASELINE configs[1], dense cube 38.3 us): 2 (32^3 voxels, 64 KB pages,
// a 1 MB table read from memory) 43.4-44.9 us in either layout -- the table entry is a second DEPENDENT load in nearly every iteration
// of a wave, because with pages 32 voxels wide some lane of the 64 has always just crossed into another page; 4 (128^3 voxels, 4 MB
// pages, a 16 KB table): see profiles/r4_raycast_notes.md.
constexpr int kPageBits = 4;                        // a page is 16 x 16 x 16 blocks
constexpr int kPageBlocks = 1 << (3 * kPageBits);   // 4 096
constexpr int kPageVoxBits = kPageBits + 3;         // ... = 128 x 128 x 128 voxels
constexpr uint32_t kPageVoxMask = (1u << kPageVoxBits) - 1u;
constexpr uint32_t kMirrorVoxels = (uint32_t)kMirrorSide * 8u;      // voxels per side of the cube
constexpr size_t kMirrorTableCells = kMirrorCells >> (3 * kPageBits);      // 16^3
constexpr int kPageNone = -1, kPageClaiming = -2, kPageUnmappable = -3;
template <bool SHORT> struct MirrorCodec;
template <> struct MirrorCodec<true> {
  using T = int16_t;
  __host__ __device__ static bool absent(T v) { return v == (int16_t)-32768; }
  __host__ __device__ static float raw(T v) { return (float)v; }
  __host__ __device__ static T of(float rawSdf) { return (int16_t)rawSdf; }
};
template <> struct MirrorCodec<false> {
  using T = uint32_t;
  __host__ __device__ static bool absent(T v) { return v == 0xffffffffu; }
  __device__ static float raw(T v) { return __uint_as_float(v); }
  __device__ static T of(float rawSdf) { return __float_as_uint(rawSdf); }
};
__host__ __device__ inline bool mirror_covers(uint32_t ux, uint32_t uy, uint32_t uz) { return ((ux | uy | uz) >> kMirrorBits) == 0u; }
// The DENSE form's cube is sized at run time from the scene's view frustum (scene.hip: the smallest power of two of blocks that holds
// the frustum with room to move -- 128^3 blocks = 2.1 GB of int16 sdf for a 3 m frustum at 4 mm voxels, where rounds 2-5 always stored
// 256^3 = 17 GB): log2 of its side in blocks, and the same tests / cell order for that side.  (A wave-uniform shift count: scalar.)
__host__ __device__ inline int mirror_dense_bits(const AccelOrigin& org) { return -org.mMaxPages; }
__host__ __device__ inline bool mirror_dense_covers(uint32_t ux, uint32_t uy, uint32_t uz, int bits) { return ((ux | uy | uz) >> bits) == 0u; }
__host__ __device__ inline uint32_t mirror_dense_cell(uint32_t ux, uint32_t uy, uint32_t uz, int bits) { return (uz << (2 * bits)) | (uy << bits) | ux; }
// the cells of the cube in plain x-fastest order (the near bits, one byte per cell)
__host__ __device__ inline uint32_t mirror_cell(uint32_t ux, uint32_t uy, uint32_t uz) { return (uz << (2 * kMirrorBits)) | (uy << kMirrorBits) | ux; }
// page-table entry of the page that holds cube-relative block (ux, uy, uz), and the block's place inside its page
__host__ __device__ inline uint32_t mirror_table_index(uint32_t ux, uint32_t uy, uint32_t uz) {
  return ((uz >> kPageBits) << (2 * (kMirrorBits - kPageBits))) | ((uy >> kPageBits) << (kMirrorBits - kPageBits)) | (ux >> kPageBits);
}
// the same from cube-relative VOXEL coordinates (each below kMirrorVoxels when the voxel is covered), and the voxel's place in its page
__host__ __device__ inline bool mirror_covers_voxel(uint32_t vx, uint32_t vy, uint32_t vz) { return ((vx | vy | vz) >> (kMirrorBits + 3)) == 0u; }
__host__ __device__ inline uint32_t mirror_table_index_voxel(uint32_t vx, uint32_t vy, uint32_t vz) {
  return ((vz >> kPageVoxBits) << (2 * (kMirrorBits - kPageBits))) | ((vy >> kPageVoxBits) << (kMirrorBits - kPageBits)) | (vx >> kPageVoxBits);
}
// BLOCK-MAJOR inside the page: the page's blocks x-fastest, a kilobyte (512 voxels, x + 8 y + 64 z) each -- the voxels rays of one wave
// read together lie in a handful of cache lines.  (Plain voxel order over the whole page -- five instructions for an address, fixed
// neighbour distances -- was measured: 38.3 -> 42.2 us even WITHOUT any table look-up, a trilinear read then touches four lines
// instead of two and neighbouring rays' voxels spread over many more: profiles/r4_raycast_notes.md.)
__host__ __device__ inline uint32_t mirror_in_page(uint32_t vx, uint32_t vy, uint32_t vz) {
  constexpr uint32_t m = (1u << kPageBits) - 1u;
  const uint32_t blk = ((((vz >> 3) & m) << kPageBits | ((vy >> 3) & m)) << kPageBits) | ((vx >> 3) & m);
  return (blk << 9) | ((vz & 7u) << 6) | ((vy & 7u) << 3) | (vx & 7u);
}
// pool index of the voxel at place `at` of page `page`
__host__ __device__ inline size_t mirror_element(int page, uint32_t at) { return ((size_t)page << (3 * kPageVoxBits)) | at; }
// place of voxel (x, y, z) of a block relative to the block's voxel (0, 0, 0): the block's own order, x + 8 y + 64 z
__host__ __device__ inline uint32_t mirror_block_voxel(uint32_t x, uint32_t y, uint32_t z) { return (z << 6) | (y << 3) | x; }
__host__ __device__ inline uint32_t mirror_block_lin(uint32_t lin) { return lin; }

// The page of a table entry, taking one from the pool if the page has none yet (allocation paths only).  Safe between the lanes of
// one wave as well: whoever wins the exchange publishes the page before it leaves the loop body, nobody waits inside the loop for a
// lane of its own wave.  Returns the page index, or kPageUnmappable.
__device__ inline int mirror_claim_page(const AccelOrigin& org, uint32_t tIdx) {
  if (org.mMaxPages >= (int)kMirrorTableCells) {      // a pool with a page for every page of the cube (measurement set-up: ITM_MIRROR_PAGES=4096): mapped 1:1
    if (org.mTable[tIdx] != (int)tIdx) __hip_atomic_store(&org.mTable[tIdx], (int)tIdx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (int)tIdx;
  }
  int v = __hip_atomic_load(&org.mTable[tIdx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (v == kPageNone || v == kPageClaiming) {
    if (v == kPageNone) {
      const int old = atomicCAS(&org.mTable[tIdx], kPageNone, kPageClaiming);
      if (old == kPageNone) {
        const int pg = atomicAdd(org.mPages, 1);
        v = pg < org.mMaxPages ? pg : kPageUnmappable;          // (a page of the pool is all "absent" until it is handed out: see the invariant above)
        __hip_atomic_store(&org.mTable[tIdx], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else v = old;
    } else v = __hip_atomic_load(&org.mTable[tIdx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return v;
}
// Plain read of a table entry by a reader whose block is the same for every lane of the wave (the integration: one block per wave) --
// through the scalar cache.  As a vector load the compiler waits for it with s_waitcnt vmcnt(0), i.e. for every voxel run the wave has
// in flight at that point; the table is not written by any launch that reads it this way.
__device__ inline int mirror_table_entry(const AccelOrigin& org, uint32_t tIdx) {
  const int32_t* q = org.mTable + tIdx;
  int v;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(q) : "memory");
  return v;
}
// Pool index of voxel (0, 0, 0) of block (bx, by, bz) -- voxel (x, y, z) of the block lies mirror_block_voxel(x, y, z) further --; false
// when the block has no place in the mirror (outside the cube, page not mapped).
// CLAIM: map the page if it is not (allocation paths); otherwise a plain read of the table.
template <bool CLAIM>
__device__ inline bool mirror_block_base(const AccelOrigin& org, int bx, int by, int bz, size_t& base) {
  const uint32_t ux = (uint32_t)(bx - org.mx), uy = (uint32_t)(by - org.my), uz = (uint32_t)(bz - org.mz);
  if (org.mMaxPages < 0) {      // dense: the cube's blocks x-fastest, a kilobyte each
    const int bits = mirror_dense_bits(org);
    if (!mirror_dense_covers(ux, uy, uz, bits)) return false;
    base = (size_t)mirror_dense_cell(ux, uy, uz, bits) << 9;
    return true;
  }
  if (org.mMaxPages == 0 || !mirror_covers(ux, uy, uz)) return false;
  const uint32_t tIdx = mirror_table_index(ux, uy, uz);
  const int page = CLAIM ? mirror_claim_page(org, tIdx) : mirror_table_entry(org, tIdx);
  if (page < 0) return false;
  base = mirror_element(page, mirror_in_page(ux << 3, uy << 3, uz << 3));
  return true;
}

// a block has just been allocated: its voxels hold the initial value (sdf 32767 / 1.0f); called by one thread (the allocation sweep)
__device__ inline void mirror_init_block(void* __restrict__ mirror, bool floatSdf, const AccelOrigin& org, int bx, int by, int bz) {
  size_t base;
  if (!mirror || !mirror_block_base<true>(org, bx, by, bz, base)) return;
  if (floatSdf) {
    uint4* q = (uint4*)((uint32_t*)mirror + base);         // 2 KB, 16-byte aligned
    const uint4 init = make_uint4(0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u);
    for (int i = 0; i < 128; ++i) q[i] = init;
  } else {
    uint4* q = (uint4*)((int16_t*)mirror + base);          // 1 KB
    const uint4 init = make_uint4(0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu);
    for (int i = 0; i < 64; ++i) q[i] = init;
  }
}

// 4x4 column-major matrix passed to kernels by value
struct Mat4 { float m[16]; };

// Matrix4 * (x,y,z,1): per row ((m0*x + m4*y) + m8*z) + m12*1, as ORUtils/Matrix.h:115-122.
// Compiled with -ffp-contract=off, so every product and sum is rounded on its own.
struct Vec3 { float x, y, z; };
__host__ __device__ inline Vec3 transform_point(const Mat4& M, float x, float y, float z) {
  Vec3 r;
  r.x = M.m[0] * x + M.m[4] * y + M.m[8] * z + M.m[12] * 1.0f;
  r.y = M.m[1] * x + M.m[5] * y + M.m[9] * z + M.m[13] * 1.0f;
  r.z = M.m[2] * x + M.m[6] * y + M.m[10] * z + M.m[14] * 1.0f;
  return r;
}

}  // namespace itm
