// Multi-GPU groups behind the C ABI (include/nxsig.h, "multi-GPU groups").
//
// The STFT / FIR path shards with no data-path exchange (channels — the reference's vectorized axes,
// lib/nx_signal.ex:358-363 — are independent; frames are independent given their samples), so a group is little more
// than one context + stream per GPU and a shard plan.  The only collective is the optional final assembly: an RCCL
// all-gather over xGMI.  Two ways to build a group:
//   LOCAL   one process drives every GPU (the Elixir / dirty-NIF host): ncclCommInitAll, collectives of all members fused
//           between ncclGroupStart / ncclGroupEnd, one stream per device so the per-device launches overlap;
//   RANKED  one process per GPU (bench.py under a launcher): ncclCommInitRank, the ncclUniqueId published by rank 0 in a
//           file on the node.
// librccl.so (570 MB) is dlopen()ed on the first group creation only: processes that never shard never pay for it, and
// libnxsig.so has no link-time dependency on it.  No torch anywhere.
#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "nxsig_internal.h"

namespace nxsig {

// ---------------------------------------------------------------------------------------------- RCCL, loaded lazily
constexpr int kMinRccl = 21800;   // 2.18.0 (ncclGetVersion: major * 10000 + minor * 100 + patch)
struct Rccl {
  void* handle = nullptr;
  std::string error, path;
  int version = 0;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

static Rccl& rccl_state() {
  static Rccl r;
  return r;
}
static Rccl* rccl() {
  Rccl& r = rccl_state();
  static std::once_flag once;
  std::call_once(once, [&r] {
    // Which librccl?  NXSIG_RCCL_LIB if set; else the ROCm installation's own (/opt/rocm/lib: the one whose headers this file was compiled
    // against) BEFORE the bare soname — inside a Python process the soname resolves to whatever LD_LIBRARY_PATH / an earlier import put
    // first (on the round-5 driver box: torch's bundled 2.26.6 instead of ROCm's 2.27.7); the bare names stay as the fallback.
    const char* names[] = {std::getenv("NXSIG_RCCL_LIB"), "/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
    for (const char* n : names) {
      if (!n || !*n) continue;
      r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (r.handle) break;
      r.error = dlerror();
    }
    if (!r.handle) return;
    bool ok = true;
    auto sym = [&](const char* name) -> void* {
      void* p = dlsym(r.handle, name);
      if (!p) { ok = false; r.error = std::string("missing symbol ") + name; }
      return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) { dlclose(r.handle); r.handle = nullptr; return; }
    // Version gate: the ten entry points bound above have kept their prototypes, enum values (ncclInt32 = 2, ncclFloat64 = 8, ncclSum = 0,
    // ncclMax = 2) and the 128-byte ncclUniqueId since NCCL 2.2; checked here against the rccl.h of 2.27.7 (this image's /opt/rocm) and run
    // against 2.26.6 (torch's bundle, round-5 driver box).  Anything older than kMinRccl is refused with a clear message instead of failing
    // in the first collective.
    typedef ncclResult_t (*GetVersionFn)(int*);
    GetVersionFn gv = reinterpret_cast<GetVersionFn>(dlsym(r.handle, "ncclGetVersion"));
    int v = 0;
    if (!gv || gv(&v) != ncclSuccess) v = 0;
    r.version = v;
    Dl_info info;
    if (dladdr(reinterpret_cast<void*>(r.AllReduce), &info) && info.dli_fname) r.path = info.dli_fname;
    if (v < kMinRccl) {
      r.error = "librccl at " + r.path + " reports version " + std::to_string(v) + ", older than the minimum " + std::to_string(kMinRccl) +
                " (2.18.0) this library accepts; point NXSIG_RCCL_LIB at a newer one";
      dlclose(r.handle); r.handle = nullptr;
    }
  });
  return r.handle ? &r : nullptr;
}
static std::string rccl_error() { return rccl_state().error; }
// one line on stderr the first time a group gets its communicators (NXSIG_QUIET=1 silences it): which librccl, which version — the
// first thing anyone debugging a multi-GPU run asks
static void rccl_announce(int rank) {
  static std::once_flag once;
  std::call_once(once, [rank] {
    const char* q = std::getenv("NXSIG_QUIET");
    if ((q && *q && *q != '0') || rank != 0) return;
    const Rccl& r = rccl_state();
    std::fprintf(stderr, "nxsig: RCCL %d.%d.%d from %s\n", r.version / 10000, (r.version / 100) % 100, r.version % 100, r.path.c_str());
  });
}

#define NXSIG_NCCL_TRY(R, expr)                                                                              \
  do {                                                                                                       \
    ncclResult_t _r = (expr);                                                                                \
    if (_r != ncclSuccess) return set_error(NXSIG_ERR_HIP, std::string(#expr) + ": " + (R)->GetErrorString(_r)); \
  } while (0)

// ncclGroupStart ... ncclGroupEnd that is closed on EVERY way out: an early error return between the two calls would leave the
// thread's RCCL group open, and every later collective of the thread would be queued and never launched (a hang).
struct NcclBracket {
  Rccl* R;
  bool open = false;
  explicit NcclBracket(Rccl* r) : R(r) {}
  ncclResult_t start() { const ncclResult_t r = R->GroupStart(); open = r == ncclSuccess; return r; }
  ncclResult_t end() { open = false; return R->GroupEnd(); }
  ~NcclBracket() { if (open) (void)R->GroupEnd(); }
};

struct Member {
  int rank = 0;
  int device = 0;
  nxsig_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  double* cell = nullptr;  // device scratch: 2 x 64 doubles (barrier word, small reductions)
};

struct Group {
  int world = 1;
  bool ranked = false;   // one process per GPU
  bool has_rccl = false;
  std::vector<Member> m;
  std::mutex mu;
};

static hipStream_t stream_of(const Member& mb) { return reinterpret_cast<hipStream_t>(nxsig_get_stream(mb.ctx)); }

static void destroy_group(Group* g) {
  if (!g) return;
  Rccl* R = g->has_rccl ? rccl() : nullptr;
  for (auto& mb : g->m) {
    if (mb.ctx) (void)nxsig_sync(mb.ctx);
    if (mb.comm && R) { (void)hipSetDevice(mb.device); (void)R->CommDestroy(mb.comm); }
    if (mb.cell) { (void)hipSetDevice(mb.device); (void)hipFree(mb.cell); }
    if (mb.ctx) nxsig_ctx_destroy(mb.ctx);
  }
  delete g;
}

static int split(int64_t total, int parts, int index, int64_t* b, int64_t* e) {
  if (parts < 1 || index < 0 || index >= parts || total < 0)
    return set_error(NXSIG_ERR_INVALID_ARG, "shard: need parts >= 1, 0 <= index < parts, total >= 0");
  const int64_t base = total / parts, extra = total % parts;
  *b = index * base + (index < extra ? index : extra);
  *e = *b + base + (index < extra ? 1 : 0);
  return NXSIG_OK;
}

static int fir_geometry(int64_t length, int32_t taps, int32_t mode, int64_t* out_len, int64_t* start) {
  if (length < 1 || taps < 1) return set_error(NXSIG_ERR_INVALID_ARG, "fir: length and num_taps must be >= 1");
  const int64_t full = length + taps - 1;
  switch (mode) {  // lib/nx_signal/convolution.ex:300-329
    case NXSIG_CONV_FULL: *out_len = full; *start = 0; break;
    case NXSIG_CONV_SAME: *out_len = length; *start = (full - length) / 2; break;
    case NXSIG_CONV_VALID: *out_len = (length >= taps ? length - taps : taps - length) + 1; *start = (full - *out_len) / 2; break;
    default: return set_error(NXSIG_ERR_INVALID_ARG, "expected mode to be one of [:full, :same, :valid]");
  }
  return NXSIG_OK;
}

// everything a member needs to know about its part of a sharded call
struct Part {
  int64_t row0 = 0, rows = 0;      // rows of the tensor it processes
  int64_t in0 = 0, in_len = 0;     // sample span of every row it reads
  int64_t out0 = 0, out_len = 0;   // output items (frames / samples) per row it produces, first item
  int64_t out_start = 0;           // FIR: index of its first output inside the full convolution of ITS input span
};

}  // namespace nxsig

using namespace nxsig;

#define NXSIG_API_BEGIN try {
#define NXSIG_API_END                                                                   \
  }                                                                                     \
  catch (const std::bad_alloc&) { return set_error(NXSIG_ERR_OOM, "host out of memory"); } \
  catch (const std::exception& e) { return set_error(NXSIG_ERR_INVALID_ARG, std::string("internal error: ") + e.what()); } \
  catch (...) { return set_error(NXSIG_ERR_INVALID_ARG, "internal error"); }

extern "C" {

/* ------------------------------------------------------------------------------------------------ shard plans (pure) */
int nxsig_shard_range(int64_t total, int32_t parts, int32_t index, int64_t* begin, int64_t* end) {
  NXSIG_API_BEGIN
  if (!begin || !end) return set_error(NXSIG_ERR_INVALID_ARG, "shard_range: null output");
  return split(total, parts, index, begin, end);
  NXSIG_API_END
}

int nxsig_shard_frames(int64_t num_frames, int32_t frame_length, int32_t hop, int32_t parts, int32_t index, int64_t* m0,
                       int64_t* m1, int64_t* s0, int64_t* s1) {
  NXSIG_API_BEGIN
  if (!m0 || !m1 || !s0 || !s1) return set_error(NXSIG_ERR_INVALID_ARG, "shard_frames: null output");
  if (frame_length < 1 || hop < 1) return set_error(NXSIG_ERR_INVALID_ARG, "shard_frames: frame_length and hop must be >= 1");
  int rc = split(num_frames, parts, index, m0, m1);
  if (rc) return rc;
  *s0 = *m0 * hop;
  *s1 = *m1 > *m0 ? (*m1 - 1) * hop + frame_length : *s0;
  return NXSIG_OK;
  NXSIG_API_END
}

int nxsig_shard_fir(int64_t length, int32_t num_taps, int32_t mode, int32_t parts, int32_t index, int64_t* n0, int64_t* n1,
                    int64_t* s0, int64_t* s1) {
  NXSIG_API_BEGIN
  if (!n0 || !n1 || !s0 || !s1) return set_error(NXSIG_ERR_INVALID_ARG, "shard_fir: null output");
  int64_t out_len, start;
  int rc = fir_geometry(length, num_taps, mode, &out_len, &start);
  if (rc) return rc;
  if ((rc = split(out_len, parts, index, n0, n1))) return rc;
  // output n of the mode's slice is full-convolution index n + start = sum_j h[j] x[n + start - j], j < taps
  int64_t a = *n0 + start - (num_taps - 1), b = *n1 + start;
  if (a < 0) a = 0;
  if (b > length) b = length;
  if (*n1 <= *n0 || b < a) b = a;
  *s0 = a; *s1 = b;
  return NXSIG_OK;
  NXSIG_API_END
}

int nxsig_shard_istft(int64_t num_frames, int32_t frame_length, int32_t hop, int32_t parts, int32_t index, int64_t* f0, int64_t* f1,
                      int64_t* n0, int64_t* n1) {
  NXSIG_API_BEGIN
  if (!f0 || !f1 || !n0 || !n1) return set_error(NXSIG_ERR_INVALID_ARG, "shard_istft: null output");
  if (frame_length < 1 || hop < 1 || hop > frame_length) return set_error(NXSIG_ERR_INVALID_ARG, "shard_istft: 1 <= hop <= frame_length required");
  if (num_frames < 1) return set_error(NXSIG_ERR_INVALID_ARG, "shard_istft: num_frames must be >= 1");
  int64_t m0, m1;
  int rc = split(num_frames, parts, index, &m0, &m1);
  if (rc) return rc;
  const int64_t R = (frame_length + hop - 1) / hop;  // frames covering one output sample
  // output samples [m0 hop, m1 hop) (the last member also takes the tail); sample n is covered by frames (n / hop - R, n / hop]
  *n0 = m0 * hop;
  *n1 = index == parts - 1 ? num_frames * hop + (frame_length - hop) : m1 * hop;
  if (m1 <= m0 && index != parts - 1) { *f0 = *f1 = m0; *n1 = *n0; return NXSIG_OK; }
  // the tuned kernels transform 2 / 4 / 8 consecutive frames in one complex FFT, groups starting at multiples of the group size:
  // ranges aligned to 8 frames see every frame in the same group position as the unsharded call (identical rounding)
  *f0 = m0 - (R - 1) > 0 ? ((m0 - (R - 1)) / 8) * 8 : 0;
  *f1 = index == parts - 1 ? num_frames : ((m1 + 7) / 8) * 8;
  if (*f1 > num_frames) *f1 = num_frames;
  if (*f1 <= *f0) { *f1 = *f0; *n1 = *n0; }
  return NXSIG_OK;
  NXSIG_API_END
}

/* ------------------------------------------------------------------------------------------------ file rendezvous */
int nxsig_rendezvous_publish(const char* path, const void* data, size_t bytes) {
  NXSIG_API_BEGIN
  if (!path || !*path || (!data && bytes)) return set_error(NXSIG_ERR_INVALID_ARG, "rendezvous_publish: bad arguments");
  // a file left behind by an earlier launch that used the same path (a crashed run of the same parent) must never be taken for
  // this launch's: drop it before anything is published.  The new content appears atomically (rename of a private temp file
  // created exclusively with mode 0600: nobody else can have planted or can read it).
  (void)::unlink(path);
  const std::string tmp = std::string(path) + ".tmp." + std::to_string((long)getpid());
  (void)::unlink(tmp.c_str());
  const int fd = ::open(tmp.c_str(), O_CREAT | O_EXCL | O_WRONLY | O_NOFOLLOW, 0600);
  if (fd < 0) return set_error(NXSIG_ERR_INVALID_ARG, "rendezvous_publish: cannot create " + tmp);
  FILE* f = ::fdopen(fd, "wb");
  if (!f) { ::close(fd); std::remove(tmp.c_str()); return set_error(NXSIG_ERR_INVALID_ARG, "rendezvous_publish: cannot open " + tmp); }
  const size_t n = bytes ? std::fwrite(data, 1, bytes, f) : 0;
  const int ce = std::fclose(f);
  if (n != bytes || ce != 0) { std::remove(tmp.c_str()); return set_error(NXSIG_ERR_INVALID_ARG, "rendezvous_publish: short write to " + tmp); }
  if (std::rename(tmp.c_str(), path) != 0) { std::remove(tmp.c_str()); return set_error(NXSIG_ERR_INVALID_ARG, std::string("rendezvous_publish: cannot rename to ") + path); }
  return NXSIG_OK;
  NXSIG_API_END
}

int nxsig_rendezvous_fetch(const char* path, void* data, size_t bytes, int32_t timeout_ms, int32_t max_age_s) {
  NXSIG_API_BEGIN
  if (!path || !*path || (!data && bytes)) return set_error(NXSIG_ERR_INVALID_ARG, "rendezvous_fetch: bad arguments");
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    // O_NOFOLLOW + fstat on the OPEN descriptor: a regular file that belongs to this user, of the expected size and young enough
    // (the fetch used to follow symlinks and never looked at the owner when the /tmp fallback directory is in use)
    const int fd = ::open(path, O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd >= 0) {
      struct stat st;
      bool ok = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_uid == geteuid() && (size_t)st.st_size == bytes &&
                (max_age_s <= 0 || std::time(nullptr) - st.st_mtime <= (time_t)max_age_s);
      size_t n = 0;
      while (ok && n < bytes) {
        const ssize_t k = ::read(fd, static_cast<char*>(data) + n, bytes - n);
        if (k <= 0) break;
        n += (size_t)k;
      }
      ::close(fd);
      if (ok && n == bytes) return NXSIG_OK;
    }
    const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
    if (ms > timeout_ms) return set_error(NXSIG_ERR_INVALID_ARG, std::string("rendezvous_fetch: timed out waiting for ") + path);
    std::this_thread::sleep_for(std::chrono::milliseconds(5));
  }
  NXSIG_API_END
}

/* ------------------------------------------------------------------------------------------------ groups */
static int make_member(Group* g, int rank, int device) {
  Member mb;
  mb.rank = rank; mb.device = device;
  int rc = nxsig_ctx_create(device, &mb.ctx);
  if (rc) return rc;
  g->m.push_back(mb);
  Member& r = g->m.back();
  NXSIG_HIP_TRY(hipSetDevice(device));
  NXSIG_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&r.cell), 128 * sizeof(double)));
  NXSIG_HIP_TRY(hipMemset(r.cell, 0, 128 * sizeof(double)));
  return NXSIG_OK;
}

int nxsig_group_create_local(int32_t n, const int32_t* device_ids, nxsig_group** out) {
  NXSIG_API_BEGIN
  if (!out) return set_error(NXSIG_ERR_INVALID_ARG, "group_create_local: out is null");
  if (n < 1 || n > 64) return set_error(NXSIG_ERR_INVALID_ARG, "group_create_local: need 1 <= n <= 64 members");
  Group* g = new Group();
  g->world = n; g->ranked = false;
  std::vector<int> devs(n);
  std::set<int> distinct;
  for (int i = 0; i < n; ++i) { devs[i] = device_ids ? device_ids[i] : i; distinct.insert(devs[i]); }
  for (int i = 0; i < n; ++i) {
    int rc = make_member(g, i, devs[i]);
    if (rc) { destroy_group(g); return rc; }
  }
  if ((int)distinct.size() == n) {  // one GPU per member: RCCL communicators (members sharing a device assemble by copies)
    Rccl* R = rccl();
    if (!R) { destroy_group(g); return set_error(NXSIG_ERR_HIP, "cannot load librccl: " + rccl_error()); }
    std::vector<ncclComm_t> comms(n, nullptr);
    ncclResult_t r = R->CommInitAll(comms.data(), n, devs.data());
    if (r != ncclSuccess) { destroy_group(g); return set_error(NXSIG_ERR_HIP, std::string("ncclCommInitAll: ") + R->GetErrorString(r)); }
    for (int i = 0; i < n; ++i) g->m[i].comm = comms[i];
    g->has_rccl = true;
    rccl_announce(0);
  }
  *out = reinterpret_cast<nxsig_group*>(g);
  return NXSIG_OK;
  NXSIG_API_END
}

int nxsig_group_create_rank(int32_t world, int32_t rank, int32_t device, const char* rendezvous_path, int32_t timeout_ms,
                            nxsig_group** out) {
  NXSIG_API_BEGIN
  if (!out) return set_error(NXSIG_ERR_INVALID_ARG, "group_create_rank: out is null");
  if (world < 1 || rank < 0 || rank >= world) return set_error(NXSIG_ERR_INVALID_ARG, "group_create_rank: need 0 <= rank < world");
  if (world > 1 && (!rendezvous_path || !*rendezvous_path))
    return set_error(NXSIG_ERR_INVALID_ARG, "group_create_rank: world > 1 needs a rendezvous path");
  Rccl* R = rccl();
  if (!R) return set_error(NXSIG_ERR_HIP, "cannot load librccl: " + rccl_error());
  Group* g = new Group();
  g->world = world; g->ranked = true;
  int rc = make_member(g, rank, device);
  if (rc) { destroy_group(g); return rc; }
  ncclUniqueId id;
  std::memset(&id, 0, sizeof(id));
  // what travels through the file: the id framed by a tag and the world size, so that a file of the right length that is not
  // this launch's id (another tool's, a different world size after a relaunch) is not accepted as one
  struct RdzvBlob { char magic[8]; int32_t world; int32_t version; ncclUniqueId id; } blob;
  static const char kMagic[8] = {'N', 'X', 'S', 'I', 'G', 'R', 'V', '1'};
  if (rank == 0) {
    ncclResult_t r = R->GetUniqueId(&id);
    if (r != ncclSuccess) { destroy_group(g); return set_error(NXSIG_ERR_HIP, std::string("ncclGetUniqueId: ") + R->GetErrorString(r)); }
    std::memset(&blob, 0, sizeof(blob));
    std::memcpy(blob.magic, kMagic, 8); blob.world = world; blob.version = 1; blob.id = id;
    if (world > 1 && (rc = nxsig_rendezvous_publish(rendezvous_path, &blob, sizeof(blob)))) { destroy_group(g); return rc; }
  } else {
    const auto t0 = std::chrono::steady_clock::now();
    const int64_t budget = timeout_ms > 0 ? timeout_ms : 120000;
    for (;;) {
      const int64_t spent = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
      if (spent >= budget) { destroy_group(g); return set_error(NXSIG_ERR_INVALID_ARG, std::string("group_create_rank: no valid rendezvous file at ") + rendezvous_path); }
      if ((rc = nxsig_rendezvous_fetch(rendezvous_path, &blob, sizeof(blob), (int32_t)(budget - spent), 600))) { destroy_group(g); return rc; }
      if (std::memcmp(blob.magic, kMagic, 8) == 0 && blob.world == world && blob.version == 1) break;
      std::this_thread::sleep_for(std::chrono::milliseconds(20));   // not ours (yet): rank 0 replaces it when it publishes
    }
    id = blob.id;
  }
  if (hipSetDevice(device) != hipSuccess) { destroy_group(g); return set_error(NXSIG_ERR_HIP, "hipSetDevice failed"); }
  ncclResult_t r = R->CommInitRank(&g->m[0].comm, world, id, rank);
  if (r != ncclSuccess) { destroy_group(g); return set_error(NXSIG_ERR_HIP, std::string("ncclCommInitRank: ") + R->GetErrorString(r)); }
  g->has_rccl = true;
  rccl_announce(rank);
  *out = reinterpret_cast<nxsig_group*>(g);
  // every rank holds the id once the communicator is up (ncclCommInitRank synchronises): rank 0 removes the file
  rc = nxsig_group_barrier(*out);
  if (rc) { destroy_group(g); *out = nullptr; return rc; }
  if (rank == 0 && world > 1) std::remove(rendezvous_path);
  return NXSIG_OK;
  NXSIG_API_END
}

void nxsig_group_destroy(nxsig_group* g) {
  try { destroy_group(reinterpret_cast<Group*>(g)); } catch (...) {}
}

int32_t nxsig_group_world(const nxsig_group* g) { return g ? reinterpret_cast<const Group*>(g)->world : 0; }
int32_t nxsig_group_local_count(const nxsig_group* g) { return g ? (int32_t)reinterpret_cast<const Group*>(g)->m.size() : 0; }
int32_t nxsig_group_rank(const nxsig_group* g, int32_t i) {
  const Group* G = reinterpret_cast<const Group*>(g);
  return (G && i >= 0 && i < (int)G->m.size()) ? G->m[i].rank : -1;
}
nxsig_ctx* nxsig_group_ctx(nxsig_group* g, int32_t i) {
  Group* G = reinterpret_cast<Group*>(g);
  return (G && i >= 0 && i < (int)G->m.size()) ? G->m[i].ctx : nullptr;
}
int nxsig_rccl_info(int32_t* version, char* path_buf, size_t buflen) {
  NXSIG_API_BEGIN
  if (version) *version = 0;
  if (path_buf && buflen) path_buf[0] = 0;
  Rccl* R = rccl();
  if (!R) return set_error(NXSIG_ERR_UNSUPPORTED, "RCCL could not be loaded: " + rccl_error());
  if (version) *version = R->version;
  if (path_buf && buflen) {
    std::strncpy(path_buf, R->path.c_str(), buflen - 1);
    path_buf[buflen - 1] = 0;
  }
  return NXSIG_OK;
  NXSIG_API_END
}

int32_t nxsig_group_has_rccl(const nxsig_group* g) { return g && reinterpret_cast<const Group*>(g)->has_rccl ? 1 : 0; }

int nxsig_group_barrier(nxsig_group* grp) {
  NXSIG_API_BEGIN
  if (!grp) return set_error(NXSIG_ERR_INVALID_ARG, "null group");
  Group* g = reinterpret_cast<Group*>(grp);
  std::lock_guard<std::mutex> lock(g->mu);
  int rc;
  for (auto& mb : g->m)
    if ((rc = nxsig_sync(mb.ctx))) return rc;
  if (!g->has_rccl) return NXSIG_OK;
  Rccl* R = rccl();
  {
    NcclBracket br(R);
    NXSIG_NCCL_TRY(R, br.start());
    for (auto& mb : g->m) {
      NXSIG_HIP_TRY(hipSetDevice(mb.device));
      int* w = reinterpret_cast<int*>(mb.cell);
      NXSIG_NCCL_TRY(R, R->AllReduce(w, w + 16, 1, ncclInt32, ncclSum, mb.comm, stream_of(mb)));
    }
    NXSIG_NCCL_TRY(R, br.end());
  }
  for (auto& mb : g->m)
    if ((rc = nxsig_sync(mb.ctx))) return rc;
  return NXSIG_OK;
  NXSIG_API_END
}

int nxsig_group_allreduce_f64(nxsig_group* grp, double* values, int32_t n, int32_t op) {
  NXSIG_API_BEGIN
  if (!grp || !values) return set_error(NXSIG_ERR_INVALID_ARG, "group_allreduce: null argument");
  if (n < 1 || n > 64) return set_error(NXSIG_ERR_INVALID_ARG, "group_allreduce: 1 <= n <= 64");
  if (op != 0 && op != 1) return set_error(NXSIG_ERR_INVALID_ARG, "group_allreduce: op must be 0 (max) or 1 (sum)");
  Group* g = reinterpret_cast<Group*>(grp);
  std::lock_guard<std::mutex> lock(g->mu);
  if (!g->has_rccl || !g->ranked) return NXSIG_OK;  // a single process: its values are the result
  Rccl* R = rccl();
  Member& mb = g->m[0];
  NXSIG_HIP_TRY(hipSetDevice(mb.device));
  hipStream_t s = stream_of(mb);
  NXSIG_HIP_TRY(hipMemcpyAsync(mb.cell, values, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
  NXSIG_NCCL_TRY(R, R->AllReduce(mb.cell, mb.cell + 64, (size_t)n, ncclDouble, op == 0 ? ncclMax : ncclSum, mb.comm, s));
  NXSIG_HIP_TRY(hipMemcpyAsync(values, mb.cell + 64, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
  NXSIG_HIP_TRY(hipStreamSynchronize(s));
  return NXSIG_OK;
  NXSIG_API_END
}

// Ranked groups (one process per GPU): before the ASSEMBLY collective of a call without an exchange step every process says whether
// its own part went well — one word, all-reduced (max) — so that a rank whose allocation, upload or kernel failed does not leave its
// peers waiting in an all-gather it never joins: either everybody gathers or everybody returns an error (its own, or "a peer failed").
// Local groups and groups without RCCL see only their own status.
static int agree_before_gather_locked(Group* g, int local_rc, bool* any_failed) {
  *any_failed = local_rc != NXSIG_OK;
  if (!(g->ranked && g->world > 1 && g->has_rccl)) return NXSIG_OK;
  Rccl* R = rccl();
  Member& mb = g->m[0];
  NXSIG_HIP_TRY(hipSetDevice(mb.device));
  hipStream_t s = stream_of(mb);
  double v = local_rc ? 1.0 : 0.0;
  NXSIG_HIP_TRY(hipMemcpyAsync(mb.cell + 96, &v, sizeof(double), hipMemcpyHostToDevice, s));
  NXSIG_NCCL_TRY(R, R->AllReduce(mb.cell + 96, mb.cell + 97, 1, ncclDouble, ncclMax, mb.comm, s));
  NXSIG_HIP_TRY(hipMemcpyAsync(&v, mb.cell + 97, sizeof(double), hipMemcpyDeviceToHost, s));
  NXSIG_HIP_TRY(hipStreamSynchronize(s));
  *any_failed = v != 0.0;
  return NXSIG_OK;
}

static int allgather_locked(Group* g, const void* const* send, const int64_t* counts, void* const* recv) {
  const int W = g->world;
  std::vector<int64_t> off(W + 1, 0);
  bool equal = true;
  for (int r = 0; r < W; ++r) {
    if (counts[r] < 0) return set_error(NXSIG_ERR_INVALID_ARG, "group_allgather: negative count");
    off[r + 1] = off[r] + counts[r];
    if (counts[r] != counts[0]) equal = false;
  }
  for (size_t i = 0; i < g->m.size(); ++i)
    if (!recv[i] || (!send[i] && counts[g->m[i].rank] > 0)) return set_error(NXSIG_ERR_INVALID_ARG, "group_allgather: null buffer");
  if (g->has_rccl) {
    Rccl* R = rccl();
    NcclBracket br(R);
    NXSIG_NCCL_TRY(R, br.start());
    for (size_t i = 0; i < g->m.size(); ++i) {
      Member& mb = g->m[i];
      NXSIG_HIP_TRY(hipSetDevice(mb.device));
      char* rb = static_cast<char*>(recv[i]);
      if (equal) {
        if (counts[0] > 0) NXSIG_NCCL_TRY(R, R->AllGather(send[i], rb, (size_t)counts[0], ncclChar, mb.comm, stream_of(mb)));
      } else {
        for (int r = 0; r < W; ++r) {  // unequal shards: one broadcast per rank, fused into a single group launch
          if (counts[r] == 0) continue;
          const void* sb = r == mb.rank ? send[i] : rb + off[r];
          NXSIG_NCCL_TRY(R, R->Broadcast(sb, rb + off[r], (size_t)counts[r], ncclChar, r, mb.comm, stream_of(mb)));
        }
      }
    }
    NXSIG_NCCL_TRY(R, br.end());
    return NXSIG_OK;
  }
  if (g->ranked) return set_error(NXSIG_ERR_UNSUPPORTED, "group_allgather: a ranked group without RCCL cannot assemble");
  // members share devices (single-GPU testing / replicas): device-to-device copies once every shard is complete
  int rc;
  for (auto& mb : g->m)
    if ((rc = nxsig_sync(mb.ctx))) return rc;
  for (size_t i = 0; i < g->m.size(); ++i) {
    NXSIG_HIP_TRY(hipSetDevice(g->m[i].device));
    char* rb = static_cast<char*>(recv[i]);
    for (size_t j = 0; j < g->m.size(); ++j) {
      const int r = g->m[j].rank;
      if (counts[r] == 0 || rb + off[r] == send[j]) continue;
      NXSIG_HIP_TRY(hipMemcpyAsync(rb + off[r], send[j], (size_t)counts[r], hipMemcpyDefault, stream_of(g->m[i])));
    }
  }
  return NXSIG_OK;
}

int nxsig_group_allgather(nxsig_group* grp, const void* const* send, const int64_t* counts, void* const* recv) {
  NXSIG_API_BEGIN
  if (!grp || !send || !counts || !recv) return set_error(NXSIG_ERR_INVALID_ARG, "group_allgather: null argument");
  Group* g = reinterpret_cast<Group*>(grp);
  std::lock_guard<std::mutex> lock(g->mu);
  return allgather_locked(g, send, counts, recv);
  NXSIG_API_END
}

}  // extern "C"

/* ------------------------------------------------------------------------------------------------ sharded hot path */
namespace {

// geometry of a sharded call: what every rank does
struct Plan {
  int64_t rows_total = 0, in_len_total = 0, out_items_total = 0;  // whole tensor: rows, samples per row, output items per row
  int64_t item_bytes = 0;                                          // bytes of one output item (frame: K * 8; sample: 4)
  std::vector<Part> part;                                          // per rank
  std::vector<int64_t> count;                                      // output bytes per rank
};

// Row stride of one member's DEVICE input shard.  batch_stride == 0: the shard is dense, f32[rows][in_len] — the form to use for
// frame / sample shards, whose spans differ from member to member when the frame count does not divide evenly (a single stride
// for the whole group cannot describe them: with batch > 1 the later members' rows would be read at the wrong offset).
// Otherwise batch_stride applies to every member and must cover the member's row.
static int shard_stride(int64_t batch_stride, int64_t rows, int64_t in_len, int64_t* out) {
  if (batch_stride == 0) { *out = in_len; return NXSIG_OK; }
  if (rows > 1 && batch_stride < in_len)
    return set_error(NXSIG_ERR_INVALID_ARG, "sharded: batch_stride is shorter than a member's rows (pass 0 for dense per-member shards)");
  *out = batch_stride;
  return NXSIG_OK;
}

// Assembly of frame / sample shards of a MULTI-ROW tensor (the reference's vectorised axes, lib/nx_signal.ex:358-363): member r holds
// its dense shard [rows][out_len_r] items; row b of it belongs at (b * out_items_total + out0_r) items of the full tensor on every
// member.  With RCCL: one ncclBroadcast per (rank, row) — root reads its dense shard, everybody (root included) receives in place —
// fused into group launches of at most ~1024 operations.  Without (members of one process sharing devices): one strided 2-D copy per
// (receiver, owner).  shard[i] = local member i's dense shard, full[i] = its full-size buffer.
static int assemble_rows_locked(Group* g, const Plan& pl, const void* const* shard, void* const* full) {
  const int W = g->world;
  const int64_t rows = pl.rows_total, row_bytes = pl.out_items_total * pl.item_bytes;
  for (size_t i = 0; i < g->m.size(); ++i) {
    const Part& q = pl.part[g->m[i].rank];
    if (!full[i] || (!shard[i] && q.rows > 0 && q.out_len > 0)) return set_error(NXSIG_ERR_INVALID_ARG, "sharded assembly: null buffer");
  }
  if (g->has_rccl) {
    Rccl* R = rccl();
    int64_t per_group = 1024 / (W > 0 ? W : 1);
    if (per_group < 1) per_group = 1;
    for (int64_t b0 = 0; b0 < rows; b0 += per_group) {
      const int64_t b1 = b0 + per_group < rows ? b0 + per_group : rows;
      NcclBracket br(R);
      NXSIG_NCCL_TRY(R, br.start());
      for (size_t i = 0; i < g->m.size(); ++i) {
        Member& mb = g->m[i];
        NXSIG_HIP_TRY(hipSetDevice(mb.device));
        char* fb = static_cast<char*>(full[i]);
        for (int r = 0; r < W; ++r) {
          const Part& q = pl.part[r];
          const int64_t seg = q.out_len * pl.item_bytes;
          if (seg == 0 || q.rows == 0) continue;
          for (int64_t b = b0; b < b1; ++b) {
            char* dst = fb + b * row_bytes + q.out0 * pl.item_bytes;
            const void* src = r == mb.rank ? static_cast<const void*>(static_cast<const char*>(shard[i]) + b * seg) : static_cast<const void*>(dst);
            NXSIG_NCCL_TRY(R, R->Broadcast(src, dst, (size_t)seg, ncclChar, r, mb.comm, stream_of(mb)));
          }
        }
      }
      NXSIG_NCCL_TRY(R, br.end());
    }
    return NXSIG_OK;
  }
  if (g->ranked && g->world > 1) return set_error(NXSIG_ERR_UNSUPPORTED, "sharded assembly: a ranked group without RCCL cannot assemble");
  int rc;
  for (auto& mb : g->m)
    if ((rc = nxsig_sync(mb.ctx))) return rc;
  for (size_t i = 0; i < g->m.size(); ++i) {
    NXSIG_HIP_TRY(hipSetDevice(g->m[i].device));
    char* fb = static_cast<char*>(full[i]);
    for (size_t j = 0; j < g->m.size(); ++j) {
      const Part& q = pl.part[g->m[j].rank];
      const int64_t seg = q.out_len * pl.item_bytes;
      if (seg == 0 || q.rows == 0) continue;
      NXSIG_HIP_TRY(hipMemcpy2DAsync(fb + q.out0 * pl.item_bytes, (size_t)row_bytes, shard[j], (size_t)seg, (size_t)seg, (size_t)rows,
                                     hipMemcpyDeviceToDevice, stream_of(g->m[i])));
    }
  }
  return NXSIG_OK;
}

// runs `compute(member, part, x_dev, out_dev)` for every local member; host mode stages through per-member device buffers.
// `exchange(dst, local_rc)` joins a COLLECTIVE: in device mode (the form ranked multi-process groups use) it is entered even when a
// local compute failed — the failure rides along as a status word, so that no rank is left waiting in the all-reduce — and every rank
// returns the error afterwards.
// `exchange(dst)` — optional — runs once every local member's compute is issued and before anything is assembled or downloaded:
// the place of a call's exchange step (sample-sharded FIR: the members agree on which rows hold a non-finite sample); dst[i] is
// member i's shard (nullptr when the member has none).
template <class Compute, class Exchange = std::nullptr_t>
int run_sharded(Group* g, const Plan& pl, const void* const* x, int64_t batch_stride, int32_t axis, int32_t gather,
                void* const* out, int32_t mem, Compute compute, Exchange exchange = nullptr) {
  constexpr bool kExchange = !std::is_same<Exchange, std::nullptr_t>::value;
  const size_t nl = g->m.size();
  const bool by_rows = axis == NXSIG_SHARD_CHANNELS;
  const bool by_row_segments = !by_rows && gather && pl.rows_total != 1;   // assembled frame / sample shards of a multi-row tensor
  int rc;
  if (mem == NXSIG_DEVICE) {
    std::vector<const void*> send(nl);
    std::vector<void*> tmp(nl, nullptr);   // by_row_segments: the member's dense shard (the full buffer takes it row by row)
    auto drop_tmp = [&]() {
      std::string keep = nxsig_last_error();
      for (size_t i = 0; i < nl; ++i)
        if (tmp[i]) { (void)nxsig_sync(g->m[i].ctx); (void)nxsig_free(g->m[i].ctx, tmp[i]); }
      (void)set_error(NXSIG_OK, keep);
    };
    int local_rc = NXSIG_OK;
    std::string local_msg;
    auto note = [&](int code) { if (code && !local_rc) { local_rc = code; local_msg = nxsig_last_error(); } };
    for (size_t i = 0; i < nl; ++i) {
      const Part& p = pl.part[g->m[i].rank];
      send[i] = nullptr;
      if (!out[i] || (!x[i] && p.rows > 0 && p.out_len > 0)) { note(set_error(NXSIG_ERR_INVALID_ARG, "sharded: null shard pointer")); continue; }
      char* dst = static_cast<char*>(out[i]);
      if (by_row_segments) {
        if (pl.count[g->m[i].rank] > 0) {
          if ((rc = nxsig_alloc(g->m[i].ctx, (size_t)pl.count[g->m[i].rank], &tmp[i]))) { note(rc); continue; }
        }
        dst = static_cast<char*>(tmp[i]);
      } else if (gather) dst += by_rows ? p.row0 * pl.out_items_total * pl.item_bytes : p.out0 * pl.item_bytes;
      send[i] = dst;
      // every member's device shard has its OWN row length (frame / sample shards: its span of the rows)
      int64_t stride = 0;
      if ((rc = shard_stride(batch_stride, p.rows, p.in_len, &stride))) { note(rc); continue; }
      if (p.rows > 0 && p.out_len > 0 && (rc = compute(g->m[i], p, static_cast<const float*>(x[i]), stride, dst))) note(rc);
    }
    if constexpr (kExchange) {
      std::vector<void*> dsts(nl);
      for (size_t i = 0; i < nl; ++i) {
        const Part& p = pl.part[g->m[i].rank];
        dsts[i] = (p.rows > 0 && p.out_len > 0) ? const_cast<void*>(send[i]) : nullptr;   // an empty part has nothing to read off or poison
      }
      rc = exchange(dsts, local_rc);
      if (local_rc) { drop_tmp(); return set_error(local_rc, local_msg); }
      if (rc) { drop_tmp(); return rc; }
    } else {
      if (local_rc && !gather) { drop_tmp(); return set_error(local_rc, local_msg); }
      if (gather) {   // the assembly is a collective: agree first (a failed rank used to skip it and its peers hung)
        bool any = false;
        const int rca = agree_before_gather_locked(g, local_rc, &any);
        if (local_rc) { drop_tmp(); return set_error(local_rc, local_msg); }
        if (rca) { drop_tmp(); return rca; }
        if (any) { drop_tmp(); return set_error(NXSIG_ERR_HIP, "sharded: another rank of the group failed before the assembly; nothing was gathered"); }
      }
    }
    if (!gather) return NXSIG_OK;
    if (by_row_segments) {
      rc = assemble_rows_locked(g, pl, send.data(), out);
      drop_tmp();   // waits for the member's stream before the dense shard goes back
      return rc;
    }
    return allgather_locked(g, send.data(), pl.count.data(), out);
  }
  // ---- host tensors: LOCAL groups only (one process sees the whole tensor); one thread per member keeps every GPU busy
  // RANKED groups (round 5): every process holds the WHOLE host tensor and a full-size host result; it computes its own member's part
  // — without `gather` only that part of its result is written, with it the shards are assembled over RCCL and every process
  // downloads the whole tensor.  (The reference's vectorised axes know no process boundary: lib/nx_signal.ex:358-363.)
  if (g->ranked && g->world > 1 && gather && !g->has_rccl)
    return set_error(NXSIG_ERR_UNSUPPORTED, "sharded: a ranked group without RCCL cannot assemble host tensors");
  if (!x[0] || !out[0]) return set_error(NXSIG_ERR_INVALID_ARG, "sharded: null tensor pointer");
  const float* xh = static_cast<const float*>(x[0]);
  char* oh = static_cast<char*>(out[0]);
  const int64_t full_bytes = pl.rows_total * pl.out_items_total * pl.item_bytes;
  std::vector<void*> din(nl, nullptr), dout(nl, nullptr), dtmp(nl, nullptr);
  std::vector<const void*> send(nl, nullptr);
  std::vector<int> rcs(nl, 0);
  std::vector<std::string> msgs(nl);
  auto cleanup = [&]() {
    for (size_t i = 0; i < nl; ++i) {
      if (din[i]) (void)nxsig_free(g->m[i].ctx, din[i]);
      if (dout[i]) (void)nxsig_free(g->m[i].ctx, dout[i]);
      if (dtmp[i]) (void)nxsig_free(g->m[i].ctx, dtmp[i]);
    }
  };
  auto work = [&](size_t i) {
    Member& mb = g->m[i];
    const Part& p = pl.part[mb.rank];
    auto fail = [&](int code) { rcs[i] = code; msgs[i] = nxsig_last_error(); };
    int r;
    const int64_t shard_bytes = pl.count[mb.rank];
    if ((r = nxsig_alloc(mb.ctx, (size_t)(gather ? full_bytes : shard_bytes), &dout[i]))) return fail(r);
    char* dst = static_cast<char*>(dout[i]);
    if (by_row_segments) {   // the dense shard; the full buffer receives it row by row (assemble_rows_locked)
      if (shard_bytes > 0 && (r = nxsig_alloc(mb.ctx, (size_t)shard_bytes, &dtmp[i]))) return fail(r);
      dst = static_cast<char*>(dtmp[i]);
    } else if (gather) dst += by_rows ? p.row0 * pl.out_items_total * pl.item_bytes : p.out0 * pl.item_bytes;
    send[i] = dst;
    if (p.rows == 0 || p.out_len == 0) return;
    if ((r = nxsig_alloc(mb.ctx, (size_t)(p.rows * p.in_len) * sizeof(float), &din[i]))) return fail(r);
    for (int64_t row = 0; row < p.rows; ++row)  // the member's rows / spans, packed densely on its device
      if ((r = nxsig_upload(mb.ctx, static_cast<float*>(din[i]) + row * p.in_len, xh + (p.row0 + row) * batch_stride + p.in0,
                            (size_t)p.in_len * sizeof(float)))) return fail(r);
    if ((r = compute(mb, p, static_cast<const float*>(din[i]), p.in_len, dst))) return fail(r);
  };
  auto fetch = [&](size_t i) {   // per-shard download straight into its place of the host result
    Member& mb = g->m[i];
    const Part& p = pl.part[mb.rank];
    auto fail = [&](int code) { rcs[i] = code; msgs[i] = nxsig_last_error(); };
    int r;
    const int64_t shard_bytes = pl.count[mb.rank];
    char* dst = static_cast<char*>(const_cast<void*>(send[i]));
    if (gather || rcs[i] || p.rows == 0 || p.out_len == 0) return;
    if (by_rows) {
      if ((r = nxsig_download(mb.ctx, oh + p.row0 * pl.out_items_total * pl.item_bytes, dst, (size_t)shard_bytes))) return fail(r);
    } else {
      for (int64_t row = 0; row < p.rows; ++row)
        if ((r = nxsig_download(mb.ctx, oh + (row * pl.out_items_total + p.out0) * pl.item_bytes, dst + row * p.out_len * pl.item_bytes,
                                (size_t)(p.out_len * pl.item_bytes)))) return fail(r);
    }
  };
  auto on_every_member = [&](auto&& fn) {
    std::vector<std::thread> th;
    for (size_t i = 1; i < nl; ++i) th.emplace_back(fn, i);
    fn(0);
    for (auto& t : th) t.join();
  };
  auto first_failure = [&]() -> int {
    for (size_t i = 0; i < nl; ++i)
      if (rcs[i]) { cleanup(); return set_error(rcs[i], msgs[i]); }
    return NXSIG_OK;
  };
  if constexpr (kExchange) {
    on_every_member(work);
    int local_rc = NXSIG_OK;
    std::string local_msg;
    for (size_t i = 0; i < nl && !local_rc; ++i)
      if (rcs[i]) { local_rc = rcs[i]; local_msg = msgs[i]; }
    {
      // the exchange is a collective of the whole group: a process whose work failed still joins it (status word), see device mode
      std::vector<void*> dsts(nl);
      for (size_t i = 0; i < nl; ++i) {
        const Part& p = pl.part[g->m[i].rank];
        dsts[i] = (!local_rc && p.rows > 0 && p.out_len > 0) ? const_cast<void*>(send[i]) : nullptr;
      }
      rc = exchange(dsts, local_rc);
      if (local_rc) { cleanup(); return set_error(local_rc, local_msg); }
      if (rc) { std::string keep = nxsig_last_error(); cleanup(); return set_error(rc, keep); }
    }
    on_every_member(fetch);
  } else {
    on_every_member([&](size_t i) { work(i); fetch(i); });
  }
  if constexpr (!kExchange) {
    if (gather) {   // every rank joins this word before anybody enters the assembly (see agree_before_gather_locked)
      int lrc = NXSIG_OK;
      for (size_t i = 0; i < nl && !lrc; ++i) lrc = rcs[i];
      bool any = false;
      const int rca = agree_before_gather_locked(g, lrc, &any);
      if ((rc = first_failure())) return rc;
      if (rca) { std::string keep = nxsig_last_error(); cleanup(); return set_error(rca, keep); }
      if (any) { cleanup(); return set_error(NXSIG_ERR_HIP, "sharded: another rank of the group failed before the assembly; nothing was gathered"); }
    }
  }
  if ((rc = first_failure())) return rc;
  if (gather) {
    rc = by_row_segments ? assemble_rows_locked(g, pl, send.data(), dout.data()) : allgather_locked(g, send.data(), pl.count.data(), dout.data());
    if (!rc) rc = nxsig_download(g->m[0].ctx, oh, dout[0], (size_t)full_bytes);  // synchronises member 0's stream
    for (size_t i = 1; i < nl && !rc; ++i) rc = nxsig_sync(g->m[i].ctx);
  }
  std::string keep = rc ? nxsig_last_error() : "";
  cleanup();
  return rc ? set_error(rc, keep) : NXSIG_OK;
}

}  // namespace

extern "C" {

int nxsig_stft_sharded_f32(nxsig_group* grp, const float* const* x, int64_t length, int32_t batch, int64_t batch_stride,
                           const float* window, const nxsig_stft_params* p, int32_t axis, int32_t gather,
                           nxsig_c64* const* z, int32_t mem) {
  NXSIG_API_BEGIN
  if (!grp || !x || !window || !p || !z) return set_error(NXSIG_ERR_INVALID_ARG, "stft_sharded: null argument");
  if (mem != NXSIG_HOST && mem != NXSIG_DEVICE) return set_error(NXSIG_ERR_INVALID_ARG, "mem must be NXSIG_HOST or NXSIG_DEVICE");
  if (axis != NXSIG_SHARD_CHANNELS && axis != NXSIG_SHARD_FRAMES) return set_error(NXSIG_ERR_INVALID_ARG, "stft_sharded: bad axis");
  if (p->pad_mode != NXSIG_PAD_VALID)
    return set_error(NXSIG_ERR_INVALID_ARG, "stft_sharded: window_padding must be :valid (padding belongs to the stream ends)");
  if (batch < 1) return set_error(NXSIG_ERR_INVALID_ARG, "stft_sharded: batch must be >= 1");
  if (mem == NXSIG_HOST && batch_stride < length) return set_error(NXSIG_ERR_INVALID_ARG, "stft_sharded: batch_stride < length");
  const int64_t M = nxsig_num_frames(length, p->frame_length, p->hop, NXSIG_PAD_VALID, 0, 0);
  if (M < 0) return (int)M;
  if (p->fft_length < 1) return set_error(NXSIG_ERR_INVALID_ARG, "stft: fft_length must be >= 1");
  Group* g = reinterpret_cast<Group*>(grp);
  std::lock_guard<std::mutex> lock(g->mu);
  Plan pl;
  pl.rows_total = batch; pl.in_len_total = length; pl.out_items_total = M; pl.item_bytes = (int64_t)p->fft_length * 8;
  pl.part.resize(g->world); pl.count.resize(g->world);
  for (int r = 0; r < g->world; ++r) {
    Part& q = pl.part[r];
    int rc;
    if (axis == NXSIG_SHARD_CHANNELS) {
      int64_t c0, c1;
      if ((rc = split(batch, g->world, r, &c0, &c1))) return rc;
      q.row0 = c0; q.rows = c1 - c0; q.in0 = 0; q.in_len = length; q.out0 = 0; q.out_len = M;
    } else {
      int64_t m0, m1, s0, s1;
      if ((rc = nxsig_shard_frames(M, p->frame_length, p->hop, g->world, r, &m0, &m1, &s0, &s1))) return rc;
      q.row0 = 0; q.rows = batch; q.in0 = s0; q.in_len = s1 - s0; q.out0 = m0; q.out_len = m1 - m0;
    }
    pl.count[r] = q.rows * q.out_len * pl.item_bytes;
  }
  auto compute = [&](Member& mb, const Part& q, const float* xd, int64_t stride, void* dst) -> int {
    return nxsig_stft_f32(mb.ctx, xd, q.in_len, (int32_t)q.rows, stride, window, p, static_cast<nxsig_c64*>(dst), nullptr, NXSIG_DEVICE);
  };
  return run_sharded(g, pl, reinterpret_cast<const void* const*>(x), batch_stride, axis, gather, reinterpret_cast<void* const*>(z), mem, compute);
  NXSIG_API_END
}

int nxsig_istft_sharded_c64(nxsig_group* grp, const nxsig_c64* const* z, int64_t num_frames, int32_t batch, const float* window,
                             const nxsig_stft_params* p, int32_t axis, int32_t gather, nxsig_c64* const* y, int32_t mem) {
  NXSIG_API_BEGIN
  if (!grp || !z || !window || !p || !y) return set_error(NXSIG_ERR_INVALID_ARG, "istft_sharded: null argument");
  if (mem != NXSIG_HOST && mem != NXSIG_DEVICE) return set_error(NXSIG_ERR_INVALID_ARG, "mem must be NXSIG_HOST or NXSIG_DEVICE");
  if (axis != NXSIG_SHARD_CHANNELS && axis != NXSIG_SHARD_FRAMES) return set_error(NXSIG_ERR_INVALID_ARG, "istft_sharded: bad axis");
  if (batch < 1 || num_frames < 1) return set_error(NXSIG_ERR_INVALID_ARG, "istft_sharded: batch and num_frames must be >= 1");
  const int N = p->frame_length, hop = p->hop, K = p->fft_length;
  if (N < 1 || hop < 1 || hop > N || K != N) return set_error(NXSIG_ERR_INVALID_ARG, "istft_sharded: 1 <= hop <= frame_length == fft_length required");
  const int64_t out_len = nxsig_ola_length(num_frames, N, hop);
  if (out_len < 0) return (int)out_len;
  Group* g = reinterpret_cast<Group*>(grp);
  std::lock_guard<std::mutex> lock(g->mu);
  // the spectrogram rides through the generic plan as rows of 2 K floats per frame
  const int64_t fpf = (int64_t)K * 2;
  Plan pl;
  pl.rows_total = batch; pl.in_len_total = num_frames * fpf; pl.out_items_total = out_len; pl.item_bytes = 8;
  pl.part.resize(g->world); pl.count.resize(g->world);
  for (int r = 0; r < g->world; ++r) {
    Part& q = pl.part[r];
    int rc;
    if (axis == NXSIG_SHARD_CHANNELS) {
      int64_t c0, c1;
      if ((rc = split(batch, g->world, r, &c0, &c1))) return rc;
      q.row0 = c0; q.rows = c1 - c0; q.in0 = 0; q.in_len = num_frames * fpf; q.out0 = 0; q.out_len = out_len; q.out_start = 0;
    } else {
      int64_t f0, f1, n0, n1;
      if ((rc = nxsig_shard_istft(num_frames, N, hop, g->world, r, &f0, &f1, &n0, &n1))) return rc;
      q.row0 = 0; q.rows = batch; q.in0 = f0 * fpf; q.in_len = (f1 - f0) * fpf; q.out0 = n0; q.out_len = n1 - n0;
      q.out_start = n0 - f0 * hop;  // first kept sample inside the member's local overlap-add
    }
    pl.count[r] = q.rows * q.out_len * pl.item_bytes;
  }
  auto compute = [&](Member& mb, const Part& q, const float* zd, int64_t /*stride: shards are dense c64[rows][frames][K]*/, void* dst) -> int {
    const int64_t frames = q.in_len / fpf;
    if (axis == NXSIG_SHARD_CHANNELS)
      return nxsig_istft_c64(mb.ctx, reinterpret_cast<const nxsig_c64*>(zd), frames, (int32_t)q.rows, window, p, static_cast<nxsig_c64*>(dst), NXSIG_DEVICE);
    // frame range: the local overlap-add of frames [f0, f1) reproduces the global one on the kept samples (every frame that
    // touches them is present, in the same order); head and tail of the local result are partial sums and are dropped
    const int64_t local_len = nxsig_ola_length(frames, N, hop);
    void* tmp = nullptr;
    int rc = nxsig_alloc(mb.ctx, (size_t)(q.rows * local_len) * 8, &tmp);
    if (rc) return rc;
    rc = nxsig_istft_c64(mb.ctx, reinterpret_cast<const nxsig_c64*>(zd), frames, (int32_t)q.rows, window, p, static_cast<nxsig_c64*>(tmp), NXSIG_DEVICE);
    if (!rc) {
      hipError_t e = hipMemcpy2DAsync(dst, (size_t)q.out_len * 8, static_cast<const char*>(tmp) + q.out_start * 8, (size_t)local_len * 8,
                                      (size_t)q.out_len * 8, (size_t)q.rows, hipMemcpyDeviceToDevice, stream_of(mb));
      if (e != hipSuccess) rc = set_error(NXSIG_ERR_HIP, std::string("istft_sharded: ") + hipGetErrorString(e));
    }
    const int rs = nxsig_sync(mb.ctx);  // the scratch result must outlive the copy
    (void)nxsig_free(mb.ctx, tmp);
    return rc ? rc : rs;
  };
  return run_sharded(g, pl, reinterpret_cast<const void* const*>(z), num_frames * fpf, axis, gather, reinterpret_cast<void* const*>(y), mem, compute);
  NXSIG_API_END
}

int nxsig_fir_sharded_f32(nxsig_group* grp, const float* const* x, int64_t length, int32_t batch, int64_t batch_stride,
                          const float* h, int32_t num_taps, int32_t mode, int32_t axis, int32_t gather, float* const* y,
                          int32_t mem) {
  NXSIG_API_BEGIN
  if (!grp || !x || !h || !y) return set_error(NXSIG_ERR_INVALID_ARG, "fir_sharded: null argument");
  if (mem != NXSIG_HOST && mem != NXSIG_DEVICE) return set_error(NXSIG_ERR_INVALID_ARG, "mem must be NXSIG_HOST or NXSIG_DEVICE");
  if (axis != NXSIG_SHARD_CHANNELS && axis != NXSIG_SHARD_FRAMES) return set_error(NXSIG_ERR_INVALID_ARG, "fir_sharded: bad axis");
  if (batch < 1) return set_error(NXSIG_ERR_INVALID_ARG, "fir_sharded: batch must be >= 1");
  if (mem == NXSIG_HOST && batch_stride < length) return set_error(NXSIG_ERR_INVALID_ARG, "fir_sharded: batch_stride < length");
  int64_t out_len, start;
  int rc = fir_geometry(length, num_taps, mode, &out_len, &start);
  if (rc) return rc;
  Group* g = reinterpret_cast<Group*>(grp);
  std::lock_guard<std::mutex> lock(g->mu);
  Plan pl;
  pl.rows_total = batch; pl.in_len_total = length; pl.out_items_total = out_len; pl.item_bytes = 4;
  pl.part.resize(g->world); pl.count.resize(g->world);
  for (int r = 0; r < g->world; ++r) {
    Part& q = pl.part[r];
    if (axis == NXSIG_SHARD_CHANNELS) {
      int64_t c0, c1;
      if ((rc = split(batch, g->world, r, &c0, &c1))) return rc;
      q.row0 = c0; q.rows = c1 - c0; q.in0 = 0; q.in_len = length; q.out0 = 0; q.out_len = out_len; q.out_start = start;
    } else {
      int64_t n0, n1, s0, s1;
      if ((rc = nxsig_shard_fir(length, num_taps, mode, g->world, r, &n0, &n1, &s0, &s1))) return rc;
      q.row0 = 0; q.rows = batch; q.in0 = s0; q.in_len = s1 - s0; q.out0 = n0; q.out_len = n1 - n0;
      q.out_start = n0 + start - s0;  // index of output n0 inside the full convolution of the span [s0, s1)
      if (q.in_len == 0) q.out_len = 0;
    }
    pl.count[r] = q.rows * q.out_len * pl.item_bytes;
  }
  auto compute = [&](Member& mb, const Part& q, const float* xd, int64_t stride, void* dst) -> int {
    return nxsig_fir_slice_f32(mb.ctx, xd, q.in_len, (int32_t)q.rows, stride, h, num_taps, q.out_start, q.out_len, static_cast<float*>(dst), NXSIG_DEVICE);
  };
  if (axis == NXSIG_SHARD_CHANNELS)   // whole rows per member: nxsig_fir_slice_f32's own poison pass sees every sample of a row
    return run_sharded(g, pl, reinterpret_cast<const void* const*>(x), batch_stride, axis, gather, reinterpret_cast<void* const*>(y), mem, compute);
  // Sample shards: the reference filters a row by ONE transform (lib/nx_signal/convolution.ex:276-284), so an Inf / NaN anywhere in
  // a row leaves no finite output in it — but a member only sees its own span.  The exchange step: every member reads off its slice
  // which rows its own poison pass turned to NaN (one int per row), the flags are all-reduced with a maximum, and every member
  // poisons the rows any member flagged.  With it the sharded call equals the unsharded one for non-finite rows too.
  if (g->ranked && g->world > 1 && !g->has_rccl)
    return set_error(NXSIG_ERR_UNSUPPORTED, "fir_sharded: a ranked group without RCCL cannot agree on the non-finite rows of sample shards");
  // flags[0 .. batch) = "row poisoned", flags[batch] = status word: a member whose compute failed contributes 1, so that every rank
  // joins the all-reduce (nobody is left waiting in it) and every rank learns that the call failed somewhere.  Whatever happens, the
  // flags are zero again when the exchange returns (stream-ordered memset): they live in the context's scratch and the next FIR call
  // of any kind reads them.
  auto exchange = [&](const std::vector<void*>& dst, int local_rc) -> int {
    const size_t nl = g->m.size();
    std::vector<int*> flags(nl, nullptr);
    struct Clear {
      Group* g; std::vector<int*>* flags; int32_t n;
      ~Clear() {
        for (size_t i = 0; i < flags->size(); ++i)
          if ((*flags)[i] && hipSetDevice(g->m[i].device) == hipSuccess)
            (void)hipMemsetAsync((*flags)[i], 0, (size_t)n * sizeof(int), stream_of(g->m[i]));
      }
    } clear{g, &flags, batch + 1};
    int rc2;
    for (size_t i = 0; i < nl; ++i) {
      Member& mb = g->m[i];
      Ctx* c = reinterpret_cast<Ctx*>(mb.ctx);
      const Part& q = pl.part[mb.rank];
      std::lock_guard<std::mutex> cl(c->mu);
      NXSIG_HIP_TRY(hipSetDevice(mb.device));
      if ((rc2 = fir_row_flags(c, batch + 1, &flags[i]))) return rc2;
      if (local_rc) {   // this process failed somewhere: its slices say nothing; only the status word matters
        const int one = 1;
        NXSIG_HIP_TRY(hipMemsetAsync(flags[i], 0, (size_t)(batch + 1) * sizeof(int), c->stream));
        NXSIG_HIP_TRY(hipMemcpyAsync(flags[i] + batch, &one, sizeof(int), hipMemcpyHostToDevice, c->stream));
        NXSIG_HIP_TRY(hipStreamSynchronize(c->stream));   // `one` lives on this stack frame
      } else if (dst[i] && (rc2 = launch_fir_flags_from_output(c, static_cast<const float*>(dst[i]), batch, q.out_len, flags[i]))) return rc2;
    }
    if (g->has_rccl) {
      Rccl* R = rccl();
      NcclBracket br(R);
      NXSIG_NCCL_TRY(R, br.start());
      for (size_t i = 0; i < nl; ++i) {
        NXSIG_HIP_TRY(hipSetDevice(g->m[i].device));
        NXSIG_NCCL_TRY(R, R->AllReduce(flags[i], flags[i], (size_t)batch + 1, ncclInt32, ncclMax, g->m[i].comm, stream_of(g->m[i])));
      }
      NXSIG_NCCL_TRY(R, br.end());
    } else if (nl > 1) {  // members of one process sharing devices (no communicators): through the host
      std::vector<int> best((size_t)batch + 1, 0), v((size_t)batch + 1);
      for (size_t i = 0; i < nl; ++i) {
        NXSIG_HIP_TRY(hipSetDevice(g->m[i].device));
        NXSIG_HIP_TRY(hipMemcpyAsync(v.data(), flags[i], v.size() * sizeof(int), hipMemcpyDeviceToHost, stream_of(g->m[i])));
        NXSIG_HIP_TRY(hipStreamSynchronize(stream_of(g->m[i])));
        for (int32_t r = 0; r <= batch; ++r) best[r] = v[r] > best[r] ? v[r] : best[r];
      }
      for (size_t i = 0; i < nl; ++i) {
        NXSIG_HIP_TRY(hipSetDevice(g->m[i].device));
        NXSIG_HIP_TRY(hipMemcpyAsync(flags[i], best.data(), best.size() * sizeof(int), hipMemcpyHostToDevice, stream_of(g->m[i])));
        NXSIG_HIP_TRY(hipStreamSynchronize(stream_of(g->m[i])));   // `best` lives on this stack frame
      }
    }
    int failed_somewhere = 0;
    {
      NXSIG_HIP_TRY(hipSetDevice(g->m[0].device));
      NXSIG_HIP_TRY(hipMemcpyAsync(&failed_somewhere, flags[0] + batch, sizeof(int), hipMemcpyDeviceToHost, stream_of(g->m[0])));
      NXSIG_HIP_TRY(hipStreamSynchronize(stream_of(g->m[0])));
    }
    if (failed_somewhere) return set_error(NXSIG_ERR_INVALID_ARG, "fir_sharded: the call failed on another member of the group");
    for (size_t i = 0; i < nl; ++i) {   // rows flagged anywhere: NaN on every member; `clear` zeroes the flags behind the pass
      Member& mb = g->m[i];
      Ctx* c = reinterpret_cast<Ctx*>(mb.ctx);
      const Part& q = pl.part[mb.rank];
      std::lock_guard<std::mutex> cl(c->mu);
      NXSIG_HIP_TRY(hipSetDevice(mb.device));
      FirLaunch f{};
      f.batch = batch; f.out_len = dst[i] ? q.out_len : 0; f.y = static_cast<float*>(dst[i]); f.row_flags = flags[i];
      if (dst[i] && (rc2 = launch_fir_poison(c, f))) return rc2;
    }
    return NXSIG_OK;
  };
  return run_sharded(g, pl, reinterpret_cast<const void* const*>(x), batch_stride, axis, gather, reinterpret_cast<void* const*>(y), mem, compute, exchange);
  NXSIG_API_END
}

int nxsig_stft_mel_sharded_f32(nxsig_group* grp, const float* const* x, int64_t length, int32_t batch, int64_t batch_stride,
                               const float* window, const nxsig_stft_params* p, int32_t mel_bins, const float* filters,
                               int32_t axis, float* const* out, int64_t* num_frames_out, int32_t mem) {
  NXSIG_API_BEGIN
  if (!grp || !x || !window || !p || !filters || !out) return set_error(NXSIG_ERR_INVALID_ARG, "stft_mel_sharded: null argument");
  if (mem != NXSIG_HOST && mem != NXSIG_DEVICE) return set_error(NXSIG_ERR_INVALID_ARG, "mem must be NXSIG_HOST or NXSIG_DEVICE");
  if (axis != NXSIG_SHARD_CHANNELS && axis != NXSIG_SHARD_FRAMES) return set_error(NXSIG_ERR_INVALID_ARG, "stft_mel_sharded: bad axis");
  if (p->pad_mode != NXSIG_PAD_VALID)
    return set_error(NXSIG_ERR_INVALID_ARG, "stft_mel_sharded: window_padding must be :valid (padding belongs to the stream ends)");
  if (batch < 1 || mel_bins < 1) return set_error(NXSIG_ERR_INVALID_ARG, "stft_mel_sharded: batch and mel_bins must be >= 1");
  if (mem == NXSIG_HOST && batch_stride < length) return set_error(NXSIG_ERR_INVALID_ARG, "stft_mel_sharded: batch_stride < length");
  const int64_t M = nxsig_num_frames(length, p->frame_length, p->hop, NXSIG_PAD_VALID, 0, 0);
  if (M < 0) return (int)M;
  if (num_frames_out) *num_frames_out = M;
  Group* g = reinterpret_cast<Group*>(grp);
  std::lock_guard<std::mutex> lock(g->mu);
  if (g->ranked && g->world > 1 && !g->has_rccl)
    return set_error(NXSIG_ERR_UNSUPPORTED, "stft_mel_sharded: a ranked group without RCCL cannot reduce the maximum");
  if (mem == NXSIG_HOST && g->ranked && g->world > 1)
    return set_error(NXSIG_ERR_UNSUPPORTED, "stft_mel_sharded: host tensors need a LOCAL group (one process, all GPUs)");
  if (mem == NXSIG_HOST && (!x[0] || !out[0])) return set_error(NXSIG_ERR_INVALID_ARG, "stft_mel_sharded: null tensor pointer");
  const size_t nl = g->m.size();
  struct Sh { int64_t row0 = 0, rows = 0, in0 = 0, in_len = 0, m0 = 0, frames = 0, count = 0; void* din = nullptr; void* dout = nullptr; int* cell = nullptr; };
  std::vector<Sh> sh(nl);
  int rc = NXSIG_OK;
  // host mode's per-member staging buffers go back to their allocators on EVERY exit (the HIP / RCCL error macros return early)
  struct Staging {
    Group* g; std::vector<Sh>* sh;
    ~Staging() {
      std::string keep = nxsig_last_error();   // nxsig_free resets the thread's message
      for (size_t i = 0; i < sh->size(); ++i) {
        if ((*sh)[i].din) (void)nxsig_free(g->m[i].ctx, (*sh)[i].din);
        if ((*sh)[i].dout) (void)nxsig_free(g->m[i].ctx, (*sh)[i].dout);
      }
      (void)set_error(NXSIG_OK, keep);
    }
  } staging{g, &sh};
  auto fail = [&](int code) { return code; };
  // pass 1 on every member: stft -> |.|^2 -> mel bands -> log10 of its shard, running maximum in the member's cell pair
  for (size_t i = 0; i < nl; ++i) {
    Member& mb = g->m[i];
    Sh& q = sh[i];
    if (axis == NXSIG_SHARD_CHANNELS) {
      int64_t c0, c1;
      if ((rc = split(batch, g->world, mb.rank, &c0, &c1))) return fail(rc);
      q.row0 = c0; q.rows = c1 - c0; q.in0 = 0; q.in_len = length; q.m0 = 0; q.frames = M;
    } else {
      int64_t m0, m1, s0, s1;
      if ((rc = nxsig_shard_frames(M, p->frame_length, p->hop, g->world, mb.rank, &m0, &m1, &s0, &s1))) return fail(rc);
      q.row0 = 0; q.rows = batch; q.in0 = s0; q.in_len = s1 - s0; q.m0 = m0; q.frames = m1 - m0;
    }
    q.count = q.rows * q.frames * mel_bins;
    Ctx* c = reinterpret_cast<Ctx*>(mb.ctx);
    NXSIG_HIP_TRY(hipSetDevice(mb.device));
    if (q.count == 0) {  // an empty shard still takes part in the reduction: identity elements
      std::lock_guard<std::mutex> cl(c->mu);
      if ((rc = launch_mel_init(c, &q.cell))) return fail(rc);
      continue;
    }
    const float* xin;
    float* dst;
    int64_t stride = batch_stride;
    if (mem == NXSIG_DEVICE) {
      if (!x[i] || !out[i]) return fail(set_error(NXSIG_ERR_INVALID_ARG, "stft_mel_sharded: null shard pointer"));
      xin = x[i]; dst = out[i];
      if ((rc = shard_stride(batch_stride, q.rows, q.in_len, &stride))) return fail(rc);
    } else {  // the member's rows / spans, packed densely on its device
      if ((rc = nxsig_alloc(mb.ctx, (size_t)(q.rows * q.in_len) * sizeof(float), &q.din))) return fail(rc);
      if ((rc = nxsig_alloc(mb.ctx, (size_t)q.count * sizeof(float), &q.dout))) return fail(rc);
      for (int64_t row = 0; row < q.rows; ++row)
        if ((rc = nxsig_upload(mb.ctx, static_cast<float*>(q.din) + row * q.in_len, x[0] + (q.row0 + row) * batch_stride + q.in0,
                               (size_t)q.in_len * sizeof(float)))) return fail(rc);
      xin = static_cast<const float*>(q.din); dst = static_cast<float*>(q.dout); stride = q.in_len;
    }
    {
      MelDeferScope defer(c);   // pass 1 only: the clamp waits for the all-reduced maximum
      rc = nxsig_stft_mel_f32(mb.ctx, xin, q.in_len, (int32_t)q.rows, stride, window, p, mel_bins, filters, dst, nullptr, NXSIG_DEVICE);
    }
    if (rc) return fail(rc);
    void* gm = nullptr;
    if ((rc = ctx_scratch(c, 5, 256, &gm))) return fail(rc);
    q.cell = reinterpret_cast<int*>(gm);
    if (mem == NXSIG_DEVICE) q.dout = nullptr;
  }
  // the exchange step of the log-mel path: max over the WHOLE tensor (Nx.reduce_max, lib/nx_signal.ex:511) and the non-finite flag
  if (g->has_rccl) {
    Rccl* R = rccl();
    NcclBracket br(R);
    NXSIG_NCCL_TRY(R, br.start());
    for (size_t i = 0; i < nl; ++i) {
      NXSIG_HIP_TRY(hipSetDevice(g->m[i].device));
      NXSIG_NCCL_TRY(R, R->AllReduce(sh[i].cell, sh[i].cell, 2, ncclInt32, ncclMax, g->m[i].comm, stream_of(g->m[i])));
    }
    NXSIG_NCCL_TRY(R, br.end());
  } else if (nl > 1) {  // members of one process sharing devices (no communicators): through the host
    int best[2] = {(int)0x80000000, 0};
    for (size_t i = 0; i < nl; ++i) {
      int v[2];
      NXSIG_HIP_TRY(hipSetDevice(g->m[i].device));
      NXSIG_HIP_TRY(hipMemcpyAsync(v, sh[i].cell, sizeof(v), hipMemcpyDeviceToHost, stream_of(g->m[i])));
      NXSIG_HIP_TRY(hipStreamSynchronize(stream_of(g->m[i])));
      best[0] = v[0] > best[0] ? v[0] : best[0];
      best[1] = v[1] > best[1] ? v[1] : best[1];
    }
    for (size_t i = 0; i < nl; ++i) {
      NXSIG_HIP_TRY(hipSetDevice(g->m[i].device));
      NXSIG_HIP_TRY(hipMemcpyAsync(sh[i].cell, best, sizeof(best), hipMemcpyHostToDevice, stream_of(g->m[i])));
      NXSIG_HIP_TRY(hipStreamSynchronize(stream_of(g->m[i])));   // `best` lives on this stack frame
    }
  }
  // pass 2 on every member: max(., global max - 8), (. + 4) / 4; host tensors: the shard goes to its place of the result
  for (size_t i = 0; i < nl; ++i) {
    Sh& q = sh[i];
    if (q.count == 0) continue;
    Member& mb = g->m[i];
    Ctx* c = reinterpret_cast<Ctx*>(mb.ctx);
    float* dst = mem == NXSIG_DEVICE ? out[i] : static_cast<float*>(q.dout);
    {
      std::lock_guard<std::mutex> cl(c->mu);
      NXSIG_HIP_TRY(hipSetDevice(mb.device));
      if ((rc = launch_mel_finish(c, dst, q.count, q.cell))) return fail(rc);
    }
    if (mem == NXSIG_DEVICE) continue;
    float* oh = out[0];
    if (axis == NXSIG_SHARD_CHANNELS) {
      if ((rc = nxsig_download(mb.ctx, oh + q.row0 * M * mel_bins, dst, (size_t)q.count * sizeof(float)))) return fail(rc);
    } else {
      for (int64_t row = 0; row < q.rows; ++row)
        if ((rc = nxsig_download(mb.ctx, oh + (row * M + q.m0) * mel_bins, dst + row * q.frames * mel_bins,
                                 (size_t)(q.frames * mel_bins) * sizeof(float)))) return fail(rc);
    }
  }
  return NXSIG_OK;
  NXSIG_API_END
}

}  // extern "C"
