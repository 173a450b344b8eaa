// io.hip -- host-side readers / writers of the data formats either side of the path: 16-bit PGM depth,
// PPM colour and the RGB-D calibration text.  Pure host code (no kernels); lives in the library so that every
// binding gets the same parser.
//
// Reference behaviour:  Utils/FileUtils.cpp:125-421 (PNM header / data, byte order),  ITMLib/Utils/ITMCalibIO.cpp:10-101,
//                       ITMLib/Objects/ITMExtrinsics.h:32-42 (inverse of the rigid transform)
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "itm_internal.h"

namespace {

enum Fmt { FMT_UNKNOWN, FMT_MONO8, FMT_RGB8, FMT_MONO16S, FMT_MONO16U };

// Header of a PGM / PPM file as the reference's reader accepts it (Utils/FileUtils.cpp:125-172): magic "P2" / "P3" (ASCII samples) or
// "P5" / "P6" (binary), then width, height and maxval as C integers ("%i": octal and hex are accepted as there), then ONE separator
// byte.  maxval selects the sample type: up to 256 bytes; for grey images up to 32 768 signed and up to 65 536 unsigned 16-bit words.
Fmt read_pnm_header(FILE* f, int* w, int* h, bool* binary) {
  char magic[1024];
  if (fscanf(f, "%1023[^ \n\t]", magic) != 1 || magic[0] != 'P' || magic[1] == '\0' || magic[2] != '\0') return FMT_UNKNOWN;
  bool grey;
  switch (magic[1]) {
    case '2': grey = true; *binary = false; break;
    case '5': grey = true; *binary = true; break;
    case '3': grey = false; *binary = false; break;
    case '6': grey = false; *binary = true; break;
    default: return FMT_UNKNOWN;
  }
  int field[3] = {0, 0, 0};                         // width, height, maxval
  for (int& v : field)
    if (fscanf(f, "%i", &v) != 1) return FMT_UNKNOWN;
  const int maxval = field[2];
  if (maxval < 0) return FMT_UNKNOWN;
  Fmt fmt = grey ? FMT_MONO8 : FMT_RGB8;
  if (maxval > (1 << 8)) {
    if (!grey || maxval > (1 << 16)) return FMT_UNKNOWN;
    fmt = maxval <= (1 << 15) ? FMT_MONO16S : FMT_MONO16U;
  }
  fgetc(f);
  *w = field[0]; *h = field[1];
  return fmt;
}

template <class T>
bool read_ascii(FILE* f, size_t n, T* dst) {
  for (size_t i = 0; i < n; ++i) { int v; if (fscanf(f, "%i", &v) != 1) return false; dst[i] = (T)v; }
  return true;
}

int fail(const std::string& m) { return itm::set_error(ITM_ERR_INVALID, m); }

}  // namespace

extern "C" {

int itm_read_depth_image(const char* path, int16_t* dst, int capacityPixels, int* w, int* h) {
  if (!path || !dst || !w || !h) return fail("null argument");
  FILE* f = fopen(path, "rb");
  if (!f) return fail(std::string("cannot open ") + path);
  bool binary; int xs, ys;
  const Fmt t = read_pnm_header(f, &xs, &ys, &binary);
  if (t != FMT_MONO16S && t != FMT_MONO16U) { fclose(f); return fail("not a 16-bit PGM depth image"); }
  const size_t n = (size_t)xs * ys;
  if (xs <= 0 || ys <= 0 || n > (size_t)capacityPixels) { fclose(f); return fail("depth image larger than the buffer"); }
  bool ok;
  if (binary) {
    ok = fread(dst, 2, n, f) == n;
    // samples are big-endian on disk
    if (ok) for (size_t i = 0; i < n; ++i) dst[i] = (int16_t)((dst[i] << 8) | ((dst[i] >> 8) & 255));
  } else ok = read_ascii(f, n, dst);
  fclose(f);
  if (!ok) return fail("truncated depth image");
  *w = xs; *h = ys;
  return ITM_OK;
}

int itm_read_rgb_image(const char* path, uint8_t* dst, int capacityPixels, int* w, int* h) {
  if (!path || !dst || !w || !h) return fail("null argument");
  FILE* f = fopen(path, "rb");
  if (!f) return fail(std::string("cannot open ") + path);
  bool binary; int xs, ys;
  const Fmt t = read_pnm_header(f, &xs, &ys, &binary);
  if (t != FMT_RGB8) { fclose(f); return fail("not an 8-bit PPM colour image"); }
  const size_t n = (size_t)xs * ys;
  if (xs <= 0 || ys <= 0 || n > (size_t)capacityPixels) { fclose(f); return fail("colour image larger than the buffer"); }
  std::vector<uint8_t> rgb(n * 3);
  const bool ok = binary ? (fread(rgb.data(), 1, n * 3, f) == n * 3) : read_ascii(f, n * 3, rgb.data());
  fclose(f);
  if (!ok) return fail("truncated colour image");
  for (size_t i = 0; i < n; ++i) { dst[4 * i] = rgb[3 * i]; dst[4 * i + 1] = rgb[3 * i + 1]; dst[4 * i + 2] = rgb[3 * i + 2]; dst[4 * i + 3] = 255; }
  *w = xs; *h = ys;
  return ITM_OK;
}

static int write_pnm(const char* path, const char* id, int maxv, int w, int h, const void* data, size_t bytes) {
  FILE* f = fopen(path, "wb");
  if (!f) return fail(std::string("cannot create ") + path);
  fprintf(f, "%s\n%i %i\n%i\n", id, w, h, maxv);
  const bool ok = fwrite(data, 1, bytes, f) == bytes;
  fclose(f);
  return ok ? ITM_OK : fail("short write");
}

int itm_write_depth_image(const char* path, const int16_t* src, int w, int h) {
  if (!path || !src || w <= 0 || h <= 0) return fail("bad argument");
  std::vector<int16_t> be((size_t)w * h);
  for (size_t i = 0; i < be.size(); ++i) be[i] = (int16_t)((src[i] << 8) | ((src[i] >> 8) & 255));
  return write_pnm(path, "P5", 65535, w, h, be.data(), be.size() * 2);
}

int itm_write_rgb_image(const char* path, const uint8_t* src, int w, int h) {
  if (!path || !src || w <= 0 || h <= 0) return fail("bad argument");
  std::vector<uint8_t> rgb((size_t)w * h * 3);
  for (size_t i = 0; i < (size_t)w * h; ++i) { rgb[3 * i] = src[4 * i]; rgb[3 * i + 1] = src[4 * i + 1]; rgb[3 * i + 2] = src[4 * i + 2]; }
  return write_pnm(path, "P6", 255, w, h, rgb.data(), rgb.size());
}

int itm_write_float_depth_image(const char* path, const float* src, int w, int h) {
  if (!path || !src || w <= 0 || h <= 0) return fail("bad argument");
  std::vector<uint16_t> mm((size_t)w * h);
  for (size_t i = 0; i < mm.size(); ++i) mm[i] = src[i] >= 0 ? (uint16_t)(src[i] * 1000.0f) : 0;   // host byte order, as the reference
  return write_pnm(path, "P5", 65535, w, h, mm.data(), mm.size() * 2);
}

int itm_read_rgbd_calib(const char* path, itm_rgbd_calib* out) {
  if (!path || !out) return fail("null argument");
  std::ifstream src(path);
  if (!src) return fail(std::string("cannot open ") + path);
  memset(out, 0, sizeof *out);
  auto intr = [&](float* size, float* k) {
    src >> size[0] >> size[1] >> k[0] >> k[1] >> k[2] >> k[3];
    return !src.fail();
  };
  if (!intr(out->size_rgb, out->intr_rgb) || !intr(out->size_d, out->intr_d)) return fail("calibration: bad intrinsics");
  float* m = out->rgb_to_depth;     // column-major: m[col * 4 + row]; the text holds rows of a 3x4 matrix
  for (int r = 0; r < 3; ++r) src >> m[0 * 4 + r] >> m[1 * 4 + r] >> m[2 * 4 + r] >> m[3 * 4 + r];
  if (src.fail()) return fail("calibration: bad extrinsics");
  m[3] = m[7] = m[11] = 0.0f; m[15] = 1.0f;
  float* inv = out->rgb_to_depth_inv;
  for (int i = 0; i < 16; ++i) inv[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) inv[r + 4 * c] = m[c + 4 * r];
  for (int r = 0; r < 3; ++r) {
    float d = 0.0f;
    for (int c = 0; c < 3; ++c) d -= m[c + 4 * r] * m[c + 4 * 3];
    inv[r + 4 * 3] = d;
  }
  std::string word;
  src >> word;
  if (src.fail()) return fail("calibration: missing disparity calibration");
  int type = 0; float a = 0, b = 0;
  if (word == "kinect") { type = 0; src >> a; }
  else if (word == "affine") { type = 1; src >> a; }
  else { std::stringstream ws(word); ws >> a; if (ws.fail()) return fail("calibration: bad disparity calibration"); }
  src >> b;
  if (src.fail()) return fail("calibration: bad disparity calibration");
  if (a == 0.0f && b == 0.0f) { type = 1; a = 1.0f / 1000.0f; b = 0.0f; }
  out->disparityType = type; out->disparityParams[0] = a; out->disparityParams[1] = b;
  return ITM_OK;
}

// ---- scene checkpoint ------------------------------------------------------------------------------------------

// One memory block per file in the layout of ORUtils/MemoryBlockPersister.h:17-130: int32 element count, raw elements.
// Written to "<file>.tmp" and renamed, so an interrupted save never leaves a half-written block under the final name.
static int save_block(const std::string& file, const void* data, size_t bytes, size_t elemBytes) {
  const std::string tmp = file + ".tmp";
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return fail("cannot create " + tmp);
  const int32_t count = (int32_t)(bytes / elemBytes);
  bool ok = fwrite(&count, 4, 1, f) == 1 && (bytes == 0 || fwrite(data, 1, bytes, f) == bytes);
  ok = (fclose(f) == 0) && ok;
  if (!ok) { remove(tmp.c_str()); return fail("short write to " + tmp); }
  if (rename(tmp.c_str(), file.c_str()) != 0) { remove(tmp.c_str()); return fail("cannot rename " + tmp); }
  return ITM_OK;
}
static int load_block(const std::string& file, void* data, size_t bytes, size_t elemBytes) {
  FILE* f = fopen(file.c_str(), "rb");
  if (!f) return fail("cannot open " + file);
  int32_t count = -1;
  bool ok = fread(&count, 4, 1, f) == 1;
  if (ok && (size_t)count != bytes / elemBytes) { fclose(f); return fail("memory block of the wrong size in " + file); }
  ok = ok && (bytes == 0 || fread(data, 1, bytes, f) == bytes);
  if (ok && fgetc(f) != EOF) { fclose(f); return fail("trailing bytes in " + file); }
  fclose(f);
  return ok ? ITM_OK : fail("truncated " + file);
}

struct BlockFile { int which; const char* name; size_t elemBytes; bool render; };

static void checkpoint_blocks(const itm_scene* s, std::vector<BlockFile>& v) {
  const bool hash = s->cfg.indexType == ITM_INDEX_HASH;
  if (hash) { v.push_back({ITM_BUF_HASH_ENTRIES, "hash.dat", 16, false}); v.push_back({ITM_BUF_EXCESS_LIST, "excess.dat", 4, false}); }
  v.push_back({ITM_BUF_ALLOCATION_LIST, "alloc.dat", 4, false});
  v.push_back({ITM_BUF_VOXEL_BLOCKS, "voxel.dat", (size_t)itm_voxel_size_bytes(s->cfg.voxelType), false});
  if (hash) { v.push_back({ITM_BUF_VISIBLE_IDS, "visible_ids.dat", 4, true}); v.push_back({ITM_BUF_VISIBLE_TYPE, "visible_type.dat", 1, true}); }
}

int itm_scene_save(const itm_scene* s, const itm_render_state* rs, const char* dir, itm_stream stream) {
  if (!s || !dir) return fail("null argument");
  if (rs && rs->scene != s) return fail("render state belongs to another scene");
  const std::string d = std::string(dir) + "/";
  std::vector<BlockFile> blocks; checkpoint_blocks(s, blocks);
  std::vector<char> host;
  // Two phases, so that a save that fails or is interrupted while the (large) blocks are being written leaves the PREVIOUS checkpoint
  // of the directory intact: every block goes to "<name>.new" first; only when all of them are on disk are they renamed over the old
  // files, config.dat last (a directory whose config.dat is older than its blocks cannot result: the window is the renames alone).
  std::vector<std::string> written;
  auto abandon = [&](int rc) { for (const std::string& f : written) remove((f + ".new").c_str()); return rc; };
  auto stage = [&](const std::string& file, const void* data, size_t bytes, size_t elemBytes) {
    const int rc = save_block(file + ".new", data, bytes, elemBytes);
    if (!rc) written.push_back(file);
    return rc;
  };
  for (const BlockFile& b : blocks) {
    if (b.render && !rs) continue;
    const size_t bytes = itm_buffer_bytes(s, rs, b.which);
    host.resize(bytes);
    int rc = itm_download(s, rs, b.which, host.data(), bytes, stream);
    if (rc) return abandon(rc);
    if ((rc = stage(d + b.name, host.data(), bytes, b.elemBytes))) return abandon(rc);
  }
  itm_counters c;
  int rc = itm_get_counters(s, rs, &c, stream);
  if (rc) return abandon(rc);
  if ((rc = stage(d + "counters.dat", &c, sizeof c, 4))) return abandon(rc);
  char cfg[sizeof(itm_scene_config) + sizeof(itm_scene_params)];
  memcpy(cfg, &s->cfg, sizeof(itm_scene_config)); memcpy(cfg + sizeof(itm_scene_config), &s->prm, sizeof(itm_scene_params));
  if ((rc = stage(d + "config.dat", cfg, sizeof cfg, 1))) return abandon(rc);
  for (const std::string& f : written)
    if (rename((f + ".new").c_str(), f.c_str()) != 0) return abandon(fail("cannot rename " + f + ".new"));
  return ITM_OK;
}

// Everything is read into host memory and validated BEFORE the first byte reaches the scene: a truncated, mismatched or
// corrupt checkpoint leaves the scene exactly as it was.  Checked: the configuration AND the scene parameters, every block's
// element count, the counters against the pool sizes, every table entry / list element / visible id against its range.
int itm_scene_load(itm_scene* s, itm_render_state* rs, const char* dir, itm_stream stream) {
  if (!s || !dir) return fail("null argument");
  if (rs && rs->scene != s) return fail("render state belongs to another scene");
  const std::string d = std::string(dir) + "/";
  char cfg[sizeof(itm_scene_config) + sizeof(itm_scene_params)];
  int rc = load_block(d + "config.dat", cfg, sizeof cfg, 1);
  if (rc) return rc;
  if (memcmp(cfg, &s->cfg, sizeof(itm_scene_config)) != 0) return fail("checkpoint was written by a scene of a different configuration");
  if (memcmp(cfg + sizeof(itm_scene_config), &s->prm, sizeof(itm_scene_params)) != 0) return fail("checkpoint was written with different scene parameters (voxelSize / mu / maxW / frustum)");
  itm_counters c;
  if ((rc = load_block(d + "counters.dat", &c, sizeof c, 4))) return rc;
  const bool hash = s->cfg.indexType == ITM_INDEX_HASH;
  const int nBlocks = hash ? s->cfg.localBlockNum : 1;
  if (hash) {
    // both counters keep decrementing once their pool is exhausted (as in the reference), so only the upper end is bounded
    if (c.lastFreeBlockId < -(1 << 30) || c.lastFreeBlockId >= nBlocks) return fail("counters.dat: lastFreeBlockId out of range");
    if (c.lastFreeExcessListId < -(1 << 30) || c.lastFreeExcessListId >= s->cfg.excessNum) return fail("counters.dat: lastFreeExcessListId out of range");
  }
  if (rs && (c.noVisibleEntries < 0 || (hash && c.noVisibleEntries > rs->capIds))) return fail("counters.dat: noVisibleEntries out of range");
  std::vector<BlockFile> blocks; checkpoint_blocks(s, blocks);
  std::vector<std::vector<char>> host(blocks.size());
  for (size_t i = 0; i < blocks.size(); ++i) {
    const BlockFile& b = blocks[i];
    if (b.render && !rs) continue;
    host[i].resize(itm_buffer_bytes(s, rs, b.which));
    if ((rc = load_block(d + b.name, host[i].data(), host[i].size(), b.elemBytes))) return rc;
    const size_t n = host[i].size() / b.elemBytes;
    if (b.which == ITM_BUF_HASH_ENTRIES) {
      const itm::HashEntry* e = (const itm::HashEntry*)host[i].data();
      for (size_t k = 0; k < n; ++k)
        if (e[k].ptr >= nBlocks || e[k].offset < 0 || e[k].offset > s->cfg.excessNum) return fail("hash.dat: entry with a pointer or chain offset outside the pools");
    } else if (b.which == ITM_BUF_EXCESS_LIST || b.which == ITM_BUF_ALLOCATION_LIST || b.which == ITM_BUF_VISIBLE_IDS) {
      const int32_t* v = (const int32_t*)host[i].data();
      const int32_t lim = b.which == ITM_BUF_EXCESS_LIST ? s->cfg.excessNum : (b.which == ITM_BUF_ALLOCATION_LIST ? nBlocks : s->noTotalEntries);
      const size_t used = (b.which == ITM_BUF_VISIBLE_IDS) ? (size_t)c.noVisibleEntries : n;     // the tail of the id list is unused
      for (size_t k = 0; k < used && k < n; ++k)
        if (v[k] < 0 || v[k] >= lim) return fail(std::string(b.name) + ": element out of range");
    } else if (b.which == ITM_BUF_VISIBLE_TYPE) {
      const unsigned char* t = (const unsigned char*)host[i].data();        // entriesVisibleType: 0 invisible, 1 / 2 visible (in memory / swapped out), 3 visible in the previous frame
      for (size_t k = 0; k < n; ++k)
        if (t[k] > 3) return fail("visible_type.dat: not a visibility type");
    }
  }
  if (rs) rs->denseRangeReady = false;        // validated: from here on the scene is being replaced
  for (size_t i = 0; i < blocks.size(); ++i) {
    const BlockFile& b = blocks[i];
    if (b.render && !rs) continue;
    if ((rc = itm_upload(s, rs, b.which, host[i].data(), host[i].size(), stream))) return rc;   // also rebuilds the occupancy bitmap / directory
  }
  return itm_set_counters(s, rs, &c, stream);
}

}  // extern "C"
