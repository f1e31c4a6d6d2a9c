// C ABI (include/i3rc_hip.h) of the gfx950 photon-tracing integrator: device-state ownership, launches, tallies.
#include "../../include/i3rc_hip.h"

#include <hip/hip_runtime.h>
#include <cstdlib>
#include <map>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "kernels.hpp"

using namespace i3rc;

namespace {

thread_local std::string g_createError;

// I3RC_POISON=1 (debugging aid): fresh device memory is filled with 0xFF bytes -- NaNs as floats, -1 as integers, wild
// as pointers -- so that a read of memory nobody wrote shows in a fresh process and not only after other
// allocations have left their contents behind.
bool poison() { static const bool on = std::getenv("I3RC_POISON") != nullptr; return on; }

// I3RC_TRACE=1 (debugging aid): a line on stderr for every fused group launched, served late or called off
bool tracing() { static const bool on = std::getenv("I3RC_TRACE") != nullptr; return on; }
double trace_ms() {
  timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
  static const double t0 = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6 - t0;
}


struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t upload(const void *src, size_t n) {
    hipError_t e = alloc(n);
    if (e != hipSuccess) return e;
    if (n) e = hipMemcpy(p, src, n, hipMemcpyHostToDevice);
    return e;
  }
  hipError_t alloc(size_t n) {
    if (p) { (void)hipFree(p); p = nullptr; }
    bytes = n;
    hipError_t e = hipMalloc(&p, n ? n : 4);
    if (e == hipSuccess && poison()) {
      // (hipMemset returns before the fill has happened, and the null stream does not order it against the library's
      // non-blocking streams: without the wait the poison lands on what the first kernels on those streams have written)
      e = hipMemset(p, 0xFF, n ? n : 4);
      if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    return e;
  }
};

}  // namespace

struct i3rc_hip_integrator {
  int device = 0;
  int nx = 0, ny = 0, nz = 0, ncomp = 0;
  std::vector<float> xE, yE, zE;  // host copies (normalisation, checks)
  DevBuf dxE, dyE, dzE, dExt, dCum, dSsa, dPf;
  DevBuf dCellRec;               // two components: a scattering's reads of its cell as one 16-byte record (DevProblem::cellRec)
  DevBuf dExtBrick;              // totalExt in bricks of 32 cells (DevProblem::extBrick)
  DevBuf dClearMap;              // ... and its clear-air map (DevProblem::clearMap)
  int clearShift = 0, clearNx = 1, clearWords = 1;
  DevBuf dColRec;                // one record per column where every column is one run of one value (DevProblem::colRec); else empty
  DevBuf dColBase;               // ... over a base profile (DevProblem::colBase, nz floats): the records then hold what lies ON the profile
  int gridPlace = I3RC_GRID_AUTO;   // test / tuning knob (i3rc_hip_select_grid_place)
  bool compDirty = true;         // comp[] changed since its device copy (dComp) was made
  int bsx = 0, bsy = 0, bsz = 0, nbx = 0, nby = 0, nbz = 0;
  std::vector<DevBuf> dInv, dInvCos, dFwd, dFwdOrig;   // per component (sized by i3rc_hip_create)
  std::vector<CompTables> comp;
  std::vector<int> nInvEntries, nFwdEntries;
  DevBuf dComp;
  DevBuf dXs, dYs, dBrdf;
  int nxs = 0, nys = 0;
  bool ldsTalliesOn = true;   // i3rc_hip_set_lds_tallies
  float brdf0 = 0.f;          // reflectance of the first surface cell (a 1 x 1 surface grid is a plain Lambertian albedo)
  DevBuf dDir;
  int nDir = 0;
  i3rc_params params{};
  float maxExt = 0.f;
  int xyRegular = 0, zRegular = 0;
  std::vector<int> maxPfIndex;
  float uniformSsa = -1.f;   // one-component domains: the value every cell shares, else -1
  bool absorbing = false;    // some cell of some component has omega < 1: launches tally volume absorption, and fluxAbsorbed is formed from it (absorbed_columns_kernel)
  int uniformPf = 0;         // ... and the phase-function entry every cell shares, else 0

  i3rc_tally_layout layout{};
  DevBuf ownTally;
  double *tally = nullptr;  // device pointer in use (own or bound)
  DevBuf workCounter;
  DevBuf srcBuf[5];

  // i3rc_hip_run_batches: batches in flight, each with a stream, a tally buffer, a work counter and a pinned host copy of its own
  struct PipeSlot {
    hipStream_t stream = nullptr;
    DevBuf tally, counter, excess;   // (excess: see accumulate_moments)
    double *pinned = nullptr;
    hipEvent_t done = nullptr;
    int batch = -1;            // batch whose tallies are on their way into `pinned`
  };
  static constexpr int kMaxInFlight = 8;
  PipeSlot pipe[kMaxInFlight];
  int64_t pipeTotal = 0;       // layout.total the slots were sized for
  // i3rc_hip_compute_batch: batches launched ahead of the caller's loop (oldest first, in slots of the same pool), all
  // with the signature below
  struct Ahead { uint32_t seed1; int slot; };
  std::vector<Ahead> aheadQueue;
  struct BatchSignature {
    uint32_t seed0 = 0; int64_t n = 0; float mu = 0.f, az = 0.f; bool set = false;
    bool operator==(const BatchSignature &o) const { return set && o.set && seed0 == o.seed0 && n == o.n && mu == o.mu && az == o.az; }
  } aheadSig, lastSig;
  uint32_t lastSeed1 = 0;

  // Fused multi-batch launches (i3rc_hip_run_batches / the look-ahead of i3rc_hip_compute_batch on problems the
  // specialised flux kernels run): a GROUP of consecutive batches is ONE grid (photon_kernel<PhiloxBatchStream, ...>), every
  // batch with tally blocks of its own -- `replicas` of them, summed by reduce_replicas_kernel into `compact` --, so that
  // one batch's tail is filled by the next batch's photons inside the launch.  Up to kFusedSlots groups are in flight, each
  // on a stream of its own (the tail of a group is covered by the next group; its copy to the host by the one after).
  struct FusedSlot {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    hipEvent_t traced = nullptr;   // the group's blocks are complete on the device (its copy to the host waits for this on the copy stream)
    DevBuf blocks, compact, counter, counterBlocks, excess;
    double *pinned = nullptr; size_t pinnedBytes = 0;
    int *abortFlag = nullptr;      // host-coherent word the kernel polls (RunArgs::abortFlag)
    int first = 0, count = 0;      // batches first .. first + count - 1 of the call (count = 0: free)
    uint32_t seed1 = 0;            // seed word of the group's first batch
    int next = 0;                  // look-ahead: batches of the group handed to the caller so far
  };
  static constexpr int kFusedSlots = 3;
  FusedSlot fused[kFusedSlots];
  // The groups' copies to the host go over a stream of their own: a copy engine does not take compute units from the next group's
  // kernel (kernels of two streams are time-sliced against each other, which is why the groups themselves share one stream), and
  // a Landsat-sized group is 100-250 MB -- 4-10 ms of PCIe during which the next group's kernel used to wait in the queue.
  hipStream_t fusedCopyStream = nullptr;
  std::vector<int> aheadGroups;    // look-ahead: slots of the groups launched ahead, oldest first
  int aheadGroupSize = 0;          // size of the next group to launch ahead (grows 8, 16, 32 ... 256)
  bool aheadBounded = false;       // i3rc_hip_expect_batches: the caller has announced its loop -- nothing is launched beyond
  uint32_t aheadEnd = 0;           // ... this seed word (exclusive)
  int fusion = -1;                 // -1: automatic, 0: never fuse, 1: fuse whatever the batch size (i3rc_hip_set_batch_fusion)
  bool fusedAheadFailed = false;   // a fused group could not be launched ahead (memory): look ahead with single batches until the layout changes
  // i3rc_hip_run_batches_moments: sums and sums of squares over a loop's batches, accumulated on the device (momArea / momDz: what
  // the normalisation needs of the grid -- column area fractions, layer depths)
  DevBuf momSum, momSq, momCounters, momArea, momDz;

  // XCD-aware photon order (launch): the sorted photon numbers and the slab bookkeeping of a launch, per stream (launches
  // on different streams -- i3rc_hip_run_batches -- are in flight together)
  struct SlabBufs { DevBuf ids, meta, blockCounts, blockBase; };
  std::map<hipStream_t, SlabBufs> slabBufs;

  hipStream_t ownStream = nullptr, stream = nullptr;
  static constexpr int kEventRing = 64;   // HIP-event pairs of the most recent timed launches
  hipEvent_t evStart[kEventRing] = {}, evStop[kEventRing] = {};
  long long timedLaunches = 0;
  int numCU = 256;
  int evThreshold = 0;        // lanes waiting before a wave runs its event phase; 0 = adapted per wave
  int lightThreshold = 0;     // lanes with an ended shadow ray before a wave runs its light phase; 0 = adapted per wave
  int blocksPerCU = 0;  // 0 = from occupancy query
  int kernelVariant = I3RC_KERNEL_AUTO;  // test / tuning knob (i3rc_hip_select_kernel)
  std::string lastKernelName;            // kernel the most recent launch ran (i3rc_hip_last_kernel_name)
  int64_t launchLimit = 0;               // photons per kernel launch (i3rc_hip_set_launch_limit); 0 = numCU * 2^22
  std::string err;

  int fail(const std::string &m) { err = m; return 1; }
  int hipfail(const char *what, hipError_t e) {
    err = std::string(what) + ": " + hipGetErrorString(e);
    return 1;
  }
};

#define HIPCHK(h, call)                                   \
  do {                                                    \
    hipError_t e__ = (call);                              \
    if (e__ != hipSuccess) return (h)->hipfail(#call, e__); \
  } while (0)

static float host_spacing(float x) {
  if (x == 0.0f) return FLT_MIN;
  int e;
  (void)std::frexp(std::fabs(x), &e);
  float r = std::ldexp(1.0f, e - 24);
  return r < FLT_MIN ? FLT_MIN : r;
}

static void compute_layout(i3rc_hip_integrator *h) {
  const int64_t ncol = (int64_t)h->nx * h->ny, ncell = ncol * h->nz;
  i3rc_tally_layout &L = h->layout;
  int64_t o = 0;
  L.fluxUp = o; o += ncol;
  L.fluxDown = o; o += ncol;
  L.fluxAbsorbed = o; o += ncol;
  L.volumeAbsorption = o; o += ncell;
  L.intensityByComponent = o; o += (int64_t)(h->ncomp + 1) * h->nDir * ncol;
  L.intensityExcess = o; o += (int64_t)(h->ncomp + 1) * h->nDir;
  L.counters = o; o += I3RC_NUM_COUNTERS;
  L.total = o;
}

static int realloc_tally(i3rc_hip_integrator *h) {
  compute_layout(h);
  HIPCHK(h, h->ownTally.alloc((size_t)h->layout.total * sizeof(double)));
  HIPCHK(h, hipMemset(h->ownTally.p, 0, (size_t)h->layout.total * sizeof(double)));
  h->tally = (double *)h->ownTally.p;
  return 0;
}

// Batches launched ahead by i3rc_hip_compute_batch read the handle's device arrays: whatever changes those (tables,
// parameters, surface, directions, tuning) waits for them first and forgets them.
static void drop_lookahead(i3rc_hip_integrator *h) {
  if (tracing() && (!h->aheadGroups.empty() || !h->aheadQueue.empty()))
    std::fprintf(stderr, "[i3rc %9.3f ms] look-ahead called off (%d fused groups, %d single batches)\n", trace_ms(), (int)h->aheadGroups.size(), (int)h->aheadQueue.size());
  for (const int k : h->aheadGroups) {   // fused groups launched ahead: called off (their waves take no further chunks), then awaited
    auto &g = h->fused[k];
    if (g.abortFlag) __atomic_store_n(g.abortFlag, 1, __ATOMIC_RELEASE);
  }
  for (const int k : h->aheadGroups) {
    auto &g = h->fused[k];
    (void)hipStreamSynchronize(g.stream);
    g.count = 0;
  }
  if (!h->aheadGroups.empty() && h->fusedCopyStream) (void)hipStreamSynchronize(h->fusedCopyStream);
  h->aheadGroups.clear();
  h->aheadGroupSize = 0;
  h->aheadBounded = false;
  for (const auto &a : h->aheadQueue) {
    (void)hipStreamSynchronize(h->pipe[a.slot].stream);
    h->pipe[a.slot].batch = -1;
  }
  h->aheadQueue.clear();
  h->aheadSig.set = false; h->lastSig.set = false;
}

// The slots of i3rc_hip_run_batches / i3rc_hip_compute_batch are kept with the handle (streams and pinned memory are
// expensive to make); a new tally layout invalidates their buffers.
static int reset_slots_if_layout_changed(i3rc_hip_integrator *h) {
  if (h->pipeTotal == h->layout.total) return 0;
  drop_lookahead(h);
  for (auto &sl : h->pipe) {
    if (sl.stream) HIPCHK(h, hipStreamSynchronize(sl.stream));
    if (sl.pinned) { HIPCHK(h, hipHostFree(sl.pinned)); sl.pinned = nullptr; }
    sl.batch = -1;
  }
  h->pipeTotal = h->layout.total;
  h->fusedAheadFailed = false;
  return 0;
}
static int ready_slot(i3rc_hip_integrator *h, int k) {
  auto &sl = h->pipe[k];
  const size_t bytes = (size_t)h->layout.total * sizeof(double);
  if (!sl.stream) HIPCHK(h, hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
  if (!sl.done) HIPCHK(h, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
  if (!sl.counter.p) HIPCHK(h, sl.counter.alloc(sizeof(unsigned long long)));
  if (!sl.pinned) HIPCHK(h, hipHostMalloc((void **)&sl.pinned, bytes, hipHostMallocDefault));
  if (sl.tally.bytes != bytes) HIPCHK(h, sl.tally.alloc(bytes));
  return 0;
}

extern "C" {

const char *i3rc_hip_version(void) { return "i3rc_hip 0.1 (gfx950)"; }

int i3rc_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return -1;
  return n;
}

const char *i3rc_hip_last_error(const i3rc_hip_integrator *h) { return h ? h->err.c_str() : g_createError.c_str(); }

int i3rc_hip_create(i3rc_hip_integrator **out, int device, int nx, int ny, int nz, int ncomp, const float *xEdges,
                    const float *yEdges, const float *zEdges, const float *totalExt, const float *cumExt, const float *ssa,
                    const int32_t *pfIndex) {
  if (!out) { g_createError = "i3rc_hip_create: null handle pointer"; return 1; }
  *out = nullptr;
  if (nx < 1 || ny < 1 || nz < 1 || ncomp < 1 || ncomp > I3RC_MAX_COMPONENTS) {
    g_createError = "i3rc_hip_create: bad dimensions (need nx,ny,nz >= 1 and 1 <= ncomp <= 255)";
    return 1;
  }
  if ((int64_t)nx * ny * nz > (int64_t)1 << 30 || (int64_t)nx * ny >= (int64_t)1 << 24 || nz >= 1 << 24) {
    g_createError = "i3rc_hip_create: domain too large (need nx*ny < 2^24, nx*ny*nz <= 2^30)";
    return 1;
  }
  if (!xEdges || !yEdges || !zEdges || !totalExt || !cumExt || !ssa || !pfIndex) {
    g_createError = "i3rc_hip_create: null array";
    return 1;
  }
  for (int i = 0; i < nx; ++i) if (!(xEdges[i + 1] > xEdges[i])) { g_createError = "i3rc_hip_create: x edges must increase"; return 1; }
  for (int i = 0; i < ny; ++i) if (!(yEdges[i + 1] > yEdges[i])) { g_createError = "i3rc_hip_create: y edges must increase"; return 1; }
  for (int i = 0; i < nz; ++i) if (!(zEdges[i + 1] > zEdges[i])) { g_createError = "i3rc_hip_create: z edges must increase"; return 1; }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev < 1) {
    g_createError = "i3rc_hip_create: no HIP device available (the integrator has no CPU fallback)";
    return 2;
  }
  if (device < 0 || device >= ndev) { g_createError = "i3rc_hip_create: device index out of range"; return 1; }

  auto *h = new i3rc_hip_integrator();
  h->device = device;
  h->nx = nx; h->ny = ny; h->nz = nz; h->ncomp = ncomp;
  h->dInv = std::vector<DevBuf>(ncomp); h->dInvCos = std::vector<DevBuf>(ncomp); h->dFwd = std::vector<DevBuf>(ncomp); h->dFwdOrig = std::vector<DevBuf>(ncomp);
  h->comp.assign(ncomp, CompTables{}); h->nInvEntries.assign(ncomp, 0); h->nFwdEntries.assign(ncomp, 0); h->maxPfIndex.assign(ncomp, 0);
  auto bail = [&](const char *what, hipError_t er) {
    g_createError = std::string(what) + ": " + hipGetErrorString(er);
    delete h;
    return 1;
  };
#define CCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return bail(#call, e_); } while (0)
  CCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  CCHK(hipGetDeviceProperties(&prop, device));
  h->numCU = prop.multiProcessorCount;
  h->xE.assign(xEdges, xEdges + nx + 1);
  h->yE.assign(yEdges, yEdges + ny + 1);
  h->zE.assign(zEdges, zEdges + nz + 1);
  const size_t ncell = (size_t)nx * ny * nz;
  CCHK(h->dxE.upload(xEdges, sizeof(float) * (nx + 1)));
  CCHK(h->dyE.upload(yEdges, sizeof(float) * (ny + 1)));
  CCHK(h->dzE.upload(zEdges, sizeof(float) * (nz + 1)));
  {
    // one layer of zeros on top: the tracer asks for the extinction of its cell before it looks at the step, and a
    // photon that starts within spacing() of the domain top has zIndex nz + 1 (it is dropped by that very step)
    std::vector<float> padded(ncell + (size_t)nx * ny, 0.0f);
    std::copy(totalExt, totalExt + ncell, padded.begin());
    CCHK(h->dExt.upload(padded.data(), sizeof(float) * padded.size()));
  }
  {
    // bricks of 32 cells: 8 deep where the grid has the layers for it (photon paths and shadow rays cross z faces
    // most often in cloud fields, whose cells are flatter than wide), the rest shared by x and y
    auto log2le = [](int n, int cap) { int s = 0; while ((2 << s) <= n && s + 1 <= cap) ++s; return s; };
    h->bsz = log2le(nz, 3);
    h->bsy = log2le(ny, (5 - h->bsz) / 2);
    if (const char *e = std::getenv("I3RC_BRICK")) {   // tuning knob: "<log2 depth>,<log2 width in y>" (the rest of the 32 cells in x)
      int bz = 3, by = 1;
      if (std::sscanf(e, "%d,%d", &bz, &by) == 2 && bz >= 0 && by >= 0 && bz + by <= 5) { h->bsz = log2le(nz, bz); h->bsy = log2le(ny, by); }
    }
    h->bsx = 5 - h->bsz - h->bsy;
    h->nbx = (nx + (1 << h->bsx) - 1) >> h->bsx; h->nby = (ny + (1 << h->bsy) - 1) >> h->bsy;
    h->nbz = (nz + 1 + (1 << h->bsz) - 1) >> h->bsz;   // (room for the layer nz + 1 of zeros, as in dExt)
    if ((int64_t)h->nbx * h->nby >= ((int64_t)1 << 24) || (int64_t)h->nbx * h->nby * h->nbz >= ((int64_t)1 << 26)) {
      g_createError = "i3rc_hip_create: domain too large for the bricked extinction copy";
      delete h;
      return 1;
    }
    std::vector<float> brick((size_t)h->nbx * h->nby * h->nbz * 32, 0.0f);
    for (int k = 0; k < nz; ++k)
      for (int j = 0; j < ny; ++j)
        for (int i = 0; i < nx; ++i) {
          const size_t b = ((size_t)(k >> h->bsz) * h->nby + (size_t)(j >> h->bsy)) * h->nbx + (size_t)(i >> h->bsx);
          const size_t w = ((size_t)(k & ((1 << h->bsz) - 1)) << (h->bsx + h->bsy)) | ((size_t)(j & ((1 << h->bsy) - 1)) << h->bsx) |
                           (size_t)(i & ((1 << h->bsx) - 1));
          brick[b * 32 + w] = totalExt[((size_t)k * ny + j) * nx + i];
        }
    CCHK(h->dExtBrick.upload(brick.data(), sizeof(float) * brick.size()));
    // clear-air map: lowest / highest layer with any extinction per footprint of 2^s x 2^s columns, at most 1024 words
    int sft = 0;
    while ((size_t)(((nx - 1) >> sft) + 1) * (size_t)(((ny - 1) >> sft) + 1) > 1024) ++sft;
    h->clearShift = sft; h->clearNx = ((nx - 1) >> sft) + 1;
    const int cny = ((ny - 1) >> sft) + 1;
    h->clearWords = h->clearNx * cny;
    std::vector<uint32_t> lo((size_t)h->clearWords, 0xffffu), hi((size_t)h->clearWords, 0u), map((size_t)h->clearWords);
    for (int k = 0; k < nz; ++k)
      for (int j = 0; j < ny; ++j)
        for (int i = 0; i < nx; ++i)
          if (totalExt[((size_t)k * ny + j) * nx + i] != 0.0f) {   // (NaN or negative values count as "something there": they are read as before)
            const size_t c = (size_t)(j >> sft) * h->clearNx + (size_t)(i >> sft);
            lo[c] = std::min<uint32_t>(lo[c], (uint32_t)std::min(k + 1, 0xfffe));
            hi[c] = std::max<uint32_t>(hi[c], (uint32_t)std::min(k + 1, 0xffff));
          }
    for (size_t c = 0; c < map.size(); ++c) map[c] = lo[c] | (hi[c] << 16);
    CCHK(h->dClearMap.upload(map.data(), sizeof(uint32_t) * map.size()));
  }
  {
    std::vector<uint32_t> rec(2 * (size_t)nx * ny);
    std::vector<float> base((size_t)nz);
    if (i3rc_hip_column_records(nx, ny, nz, totalExt, rec.data()) == 1) CCHK(h->dColRec.upload(rec.data(), sizeof(uint32_t) * rec.size()));
    else if (i3rc_hip_column_records_base(nx, ny, nz, totalExt, rec.data(), base.data()) == 1) {   // the same over a value per layer (a uniform gas under / around the clouds)
      CCHK(h->dColRec.upload(rec.data(), sizeof(uint32_t) * rec.size()));
      CCHK(h->dColBase.upload(base.data(), sizeof(float) * base.size()));
    }
  }
  CCHK(h->dCum.upload(cumExt, sizeof(float) * ncell * ncomp));
  CCHK(h->dSsa.upload(ssa, sizeof(float) * ncell * ncomp));
  CCHK(h->dPf.upload(pfIndex, sizeof(int32_t) * ncell * ncomp));
  if (ncomp == 2 || ncomp == 3) {   // (DevProblem::cellRec)
    static const bool recOn = !(std::getenv("I3RC_CELL_RECORDS") && std::atoi(std::getenv("I3RC_CELL_RECORDS")) == 0);
    bool fits = recOn;
    for (size_t i = 0; i < 2 * ncell && fits; ++i) fits = pfIndex[i] >= 0 && pfIndex[i] < 65536;   // (the first two entries share a word)
    if (fits) {
      auto bits = [](float v) { uint32_t b; std::memcpy(&b, &v, 4); return b; };
      const size_t words = ncomp == 2 ? 4 : 8;
      std::vector<uint32_t> rec(words * ncell, 0u);
      for (size_t i = 0; i < ncell; ++i) {
        uint32_t *r = &rec[words * i];
        const uint32_t pf01 = (uint32_t)pfIndex[i] | ((uint32_t)pfIndex[ncell + i] << 16);
        if (ncomp == 2) { r[0] = bits(cumExt[i]); r[1] = bits(ssa[i]); r[2] = bits(ssa[ncell + i]); r[3] = pf01; }
        else {
          r[0] = bits(cumExt[i]); r[1] = bits(cumExt[ncell + i]); r[2] = bits(ssa[i]); r[3] = bits(ssa[ncell + i]);
          r[4] = bits(ssa[2 * ncell + i]); r[5] = pf01; r[6] = (uint32_t)pfIndex[2 * ncell + i];
        }
      }
      CCHK(h->dCellRec.upload(rec.data(), sizeof(uint32_t) * rec.size()));
    }
  }
  CCHK(h->workCounter.alloc(sizeof(unsigned long long)));
  CCHK(h->dComp.alloc(sizeof(CompTables) * ncomp));
  CCHK(h->dDir.alloc(sizeof(float) * 3 * I3RC_MAX_DIRECTIONS));
  CCHK(hipStreamCreateWithFlags(&h->ownStream, hipStreamNonBlocking));
  h->stream = h->ownStream;
  for (int i = 0; i < i3rc_hip_integrator::kEventRing; ++i) {
    CCHK(hipEventCreate(&h->evStart[i]));
    CCHK(hipEventCreate(&h->evStop[i]));
  }
#undef CCHK
  // regular-spacing flags, new_Integrator :193-211
  {
    const float dx = xEdges[1] - xEdges[0], dy = yEdges[1] - yEdges[0], dz = zEdges[1] - zEdges[0];
    int xy = 1, z = 1;
    for (int i = 0; i < nx; ++i) if (!(std::fabs((xEdges[i + 1] - xEdges[i]) - dx) <= 2.0f * host_spacing(xEdges[i + 1]))) xy = 0;
    for (int i = 0; i < ny; ++i) if (!(std::fabs((yEdges[i + 1] - yEdges[i]) - dy) <= 2.0f * host_spacing(yEdges[i + 1]))) xy = 0;
    for (int i = 0; i < nz; ++i) if (!(std::fabs((zEdges[i + 1] - zEdges[i]) - dz) <= host_spacing(zEdges[i + 1]))) z = 0;
    h->xyRegular = xy; h->zRegular = z;
  }
  h->maxExt = totalExt[0];
  for (size_t i = 1; i < ncell; ++i) h->maxExt = std::max(h->maxExt, totalExt[i]);  // computeRT :438-439
  for (int c = 0; c < ncomp; ++c) {
    int m = 0;
    for (size_t i = 0; i < ncell; ++i) m = std::max(m, pfIndex[(size_t)c * ncell + i]);
    h->maxPfIndex[c] = m;
  }
  for (size_t i = 0; i < ncell * (size_t)ncomp && !h->absorbing; ++i) h->absorbing = ssa[i] < 1.0f;
  if (ncomp == 1) {
    // Values that every cell WITH EXTINCTION shares travel in the kernel arguments (specialised kernels: ray tracing, where a
    // photon can only be scattered in a cell of positive extinction -- the tracer never stops in any other --, so what the
    // clear cells hold is never read: the I3RC cloud fields have omega = 0 and phase-function entry 0 there).  Without this
    // every scattering reads two more words from two more arrays of the field's size, which on the Landsat fields
    // do not fit in L2 beside it.
    bool sameSsa = true, samePf = true, any = false;
    float ssa0 = 1.f; int32_t pf0 = 1;
    for (size_t i = 0; i < ncell && (sameSsa || samePf); ++i) {
      if (totalExt[i] == 0.0f) continue;
      if (!any) { any = true; ssa0 = ssa[i]; pf0 = pfIndex[i]; continue; }
      sameSsa = sameSsa && ssa[i] == ssa0;
      samePf = samePf && pfIndex[i] == pf0;
    }
    h->uniformSsa = (sameSsa && ssa0 >= 0.f) ? ssa0 : -1.f;
    h->uniformPf = (samePf && pf0 >= 1) ? pf0 : 0;
  }
  if (ncomp == 1 && h->uniformSsa < 0.0f && h->uniformPf < 1) {   // one component, neither albedo nor entry shared: {ssa, pfIndex}, 8 bytes a cell
    static const bool recOn1 = !(std::getenv("I3RC_CELL_RECORDS") && std::atoi(std::getenv("I3RC_CELL_RECORDS")) == 0);
    if (recOn1) {
      std::vector<uint32_t> rec(2 * ncell);
      for (size_t i = 0; i < ncell; ++i) { std::memcpy(&rec[2 * i], &ssa[i], 4); rec[2 * i + 1] = (uint32_t)pfIndex[i]; }
      if (h->dCellRec.upload(rec.data(), sizeof(uint32_t) * rec.size()) != hipSuccess) { g_createError = "i3rc_hip_create: device allocation of the cell records failed"; delete h; return 1; }
    }
  }
  // defaults of type(integrator) :54-129
  h->params.surfaceAlbedo = 0.f; h->params.useSurfaceBDRF = 0; h->params.useRayTracing = 1; h->params.useRussianRoulette = 1;
  h->params.useHybridPhaseFunsForIntenCalcs = 0; h->params.numOrdersOrigPhaseFunIntenCalcs = 0;
  h->params.useRussianRouletteForIntensity = 0; h->params.zetaMin = 0.3f; h->params.limitIntensityContributions = 0;
  h->params.maxIntensityContribution = FLT_MAX;
  if (realloc_tally(h)) { g_createError = h->err; delete h; return 1; }
  *out = h;
  return 0;
}

int i3rc_hip_destroy(i3rc_hip_integrator *h) {
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  // (first of all: whatever was launched ahead of the caller is called off -- destroying a stream waits for the device)
  for (auto &g : h->fused) if (g.abortFlag) __atomic_store_n(g.abortFlag, 1, __ATOMIC_RELEASE);
  if (tracing()) std::fprintf(stderr, "[i3rc %9.3f ms] destroy begins\n", trace_ms());
  for (auto &g : h->fused) if (g.stream && g.count > 0) { (void)hipStreamSynchronize(g.stream); break; }
  if (h->ownStream) { (void)hipStreamSynchronize(h->ownStream); (void)hipStreamDestroy(h->ownStream); }
  for (auto &sl : h->pipe) {
    if (sl.stream) { (void)hipStreamSynchronize(sl.stream); (void)hipStreamDestroy(sl.stream); }
    if (sl.done) (void)hipEventDestroy(sl.done);
    if (sl.pinned) (void)hipHostFree(sl.pinned);
  }
  for (auto &g : h->fused) if (g.abortFlag) __atomic_store_n(g.abortFlag, 1, __ATOMIC_RELEASE);
  if (tracing()) std::fprintf(stderr, "[i3rc %9.3f ms] destroy: abort words set\n", trace_ms());
  for (int k = 0; k < i3rc_hip_integrator::kFusedSlots; ++k) {
    auto &g = h->fused[k];
    bool shared = false;   // (the slots normally share the first one's stream)
    for (int j = 0; j < k; ++j) shared = shared || (g.stream && g.stream == h->fused[j].stream);
    if (g.stream && !shared) { (void)hipStreamSynchronize(g.stream); if (tracing()) std::fprintf(stderr, "[i3rc %9.3f ms] destroy: fused stream drained\n", trace_ms()); (void)hipStreamDestroy(g.stream); }
    if (g.done) (void)hipEventDestroy(g.done);
    if (g.traced) (void)hipEventDestroy(g.traced);
    if (g.pinned) (void)hipHostFree(g.pinned);
    if (g.abortFlag) (void)hipHostFree(g.abortFlag);
  }
  if (h->fusedCopyStream) { (void)hipStreamSynchronize(h->fusedCopyStream); (void)hipStreamDestroy(h->fusedCopyStream); }
  for (int i = 0; i < i3rc_hip_integrator::kEventRing; ++i) {
    if (h->evStart[i]) (void)hipEventDestroy(h->evStart[i]);
    if (h->evStop[i]) (void)hipEventDestroy(h->evStop[i]);
  }
  delete h;
  return 0;
}

int i3rc_hip_set_inverse_table(i3rc_hip_integrator *h, int comp, int nSteps, int nEntries, const float *t) {
  if (!h) return 1;
  drop_lookahead(h);
  if (comp < 1 || comp > h->ncomp) return h->fail("i3rc_hip_set_inverse_table: component out of range");
  if (nSteps < 2 || nEntries < 1 || !t) return h->fail("i3rc_hip_set_inverse_table: bad table");
  if (nEntries < h->maxPfIndex[comp - 1]) return h->fail("i3rc_hip_set_inverse_table: phaseFunctionIndex refers to a missing table entry");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, h->dInv[comp - 1].upload(t, sizeof(float) * (size_t)nSteps * nEntries));
  // cos of every tabulated angle, in float64, rounded once (see scattering_cosine in tracer.hpp)
  std::vector<float> cosTab((size_t)nSteps * nEntries);
  for (size_t i = 0; i < cosTab.size(); ++i) cosTab[i] = (float)std::cos((double)t[i]);
  HIPCHK(h, h->dInvCos[comp - 1].upload(cosTab.data(), sizeof(float) * cosTab.size()));
  h->comp[comp - 1].inv = (const float *)h->dInv[comp - 1].p;
  h->comp[comp - 1].invCos = (const float *)h->dInvCos[comp - 1].p;
  h->comp[comp - 1].nInv = nSteps;
  h->compDirty = true;
  h->nInvEntries[comp - 1] = nEntries;
  return 0;
}

int i3rc_hip_set_forward_tables(i3rc_hip_integrator *h, int comp, int nSteps, int nEntries, const float *hybrid, const float *orig) {
  if (!h) return 1;
  drop_lookahead(h);
  if (comp < 1 || comp > h->ncomp) return h->fail("i3rc_hip_set_forward_tables: component out of range");
  if (nSteps < 2 || nEntries < 1 || !hybrid) return h->fail("i3rc_hip_set_forward_tables: bad table");
  if (nEntries < h->maxPfIndex[comp - 1]) return h->fail("i3rc_hip_set_forward_tables: phaseFunctionIndex refers to a missing table entry");
  if (!orig) orig = hybrid;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, h->dFwd[comp - 1].upload(hybrid, sizeof(float) * (size_t)nSteps * nEntries));
  HIPCHK(h, h->dFwdOrig[comp - 1].upload(orig, sizeof(float) * (size_t)nSteps * nEntries));
  h->comp[comp - 1].fwd = (const float *)h->dFwd[comp - 1].p;
  h->comp[comp - 1].fwdOrig = (const float *)h->dFwdOrig[comp - 1].p;
  h->comp[comp - 1].nFwd = nSteps;
  h->compDirty = true;
  h->nFwdEntries[comp - 1] = nEntries;
  return 0;
}

int i3rc_hip_set_params(i3rc_hip_integrator *h, const i3rc_params *p) {
  if (!h) return 1;
  drop_lookahead(h);
  if (!p) return h->fail("i3rc_hip_set_params: null params");
  if (!p->useSurfaceBDRF && (p->surfaceAlbedo > 1.f || p->surfaceAlbedo < 0.f))
    return h->fail("specifyParameters: surface albedo out of range.");  // :878-879
  if (p->useSurfaceBDRF && !h->dBrdf.p) return h->fail("specifyParameters: surface description isn't valid.");  // :882-883
  if (p->zetaMin < 0.f) return h->fail("specifyParameters: zetaMin must be >= 0.");
  h->params = *p;
  return 0;
}

int i3rc_hip_set_surface(i3rc_hip_integrator *h, int nxs, int nys, const float *xs, const float *ys, const float *brdf) {
  if (!h) return 1;
  drop_lookahead(h);
  if (nxs < 1 || nys < 1 || !xs || !ys || !brdf) return h->fail("i3rc_hip_set_surface: bad surface grid");
  for (int i = 0; i < nxs; ++i) if (!(xs[i + 1] > xs[i])) return h->fail("new_SurfaceDescription: positions must be unique, increasing.");
  for (int i = 0; i < nys; ++i) if (!(ys[i + 1] > ys[i])) return h->fail("new_SurfaceDescription: positions must be unique, increasing.");
  for (size_t i = 0; i < (size_t)nxs * nys; ++i)
    if (brdf[i] < 0.f || brdf[i] > 1.f) return h->fail("new_SurfaceDescription: surface reflectance must be between 0 and 1");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, h->dXs.upload(xs, sizeof(float) * (nxs + 1)));
  HIPCHK(h, h->dYs.upload(ys, sizeof(float) * (nys + 1)));
  HIPCHK(h, h->dBrdf.upload(brdf, sizeof(float) * (size_t)nxs * nys));
  h->nxs = nxs; h->nys = nys;
  h->brdf0 = brdf[0];
  return 0;
}

int i3rc_hip_set_directions(i3rc_hip_integrator *h, int nDir, const float *dirCos) {
  if (!h) return 1;
  drop_lookahead(h);
  if (nDir < 0 || nDir > I3RC_MAX_DIRECTIONS) return h->fail("i3rc_hip_set_directions: 0 <= nDir <= 255 required");
  if (nDir > 0 && !dirCos) return h->fail("i3rc_hip_set_directions: null directions");
  for (int d = 0; d < nDir; ++d)
    if (std::fabs(dirCos[3 * d + 2]) < FLT_MIN) return h->fail("specifyParameters: intensityMus can't be 0 (directly sideways)");  // :932-933
  // A change of nDir changes the tally layout.  With a caller-bound tally buffer that is refused BEFORE anything is
  // touched: nDir, the device copy of the directions and the layout all stay as they are (the caller unbinds --
  // i3rc_hip_bind_tally_buffer(h, NULL, 0) --, sets the directions, asks for the new layout and binds a buffer of that size).
  const bool changed = nDir != h->nDir;
  if (changed && h->tally != (double *)h->ownTally.p)
    return h->fail("i3rc_hip_set_directions: a caller-bound tally buffer is in use; unbind it (bind NULL) before changing nDir, "
                   "then bind a buffer of the new layout's size");
  HIPCHK(h, hipSetDevice(h->device));
  // launches in flight on the (possibly non-blocking) stream read dDir / the tally buffer: drain them first
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (nDir > 0) HIPCHK(h, hipMemcpy(h->dDir.p, dirCos, sizeof(float) * 3 * nDir, hipMemcpyHostToDevice));
  h->nDir = nDir;
  if (changed) return realloc_tally(h);
  return 0;
}

int i3rc_hip_get_tally_layout(const i3rc_hip_integrator *h, i3rc_tally_layout *layout) {
  if (!h || !layout) return 1;
  *layout = h->layout;
  return 0;
}

int i3rc_hip_bind_tally_buffer(i3rc_hip_integrator *h, void *devicePtr, size_t bytes) {
  if (!h) return 1;
  if (!devicePtr) { h->tally = (double *)h->ownTally.p; return 0; }
  if (bytes < (size_t)h->layout.total * sizeof(double)) return h->fail("i3rc_hip_bind_tally_buffer: buffer too small");
  if (((uintptr_t)devicePtr & 7u) != 0) return h->fail("i3rc_hip_bind_tally_buffer: buffer must be 8-byte aligned");
  h->tally = (double *)devicePtr;
  return 0;
}

int i3rc_hip_set_stream(i3rc_hip_integrator *h, void *s) {
  if (!h) return 1;
  h->stream = (hipStream_t)s;   // NULL is HIP's null stream (what torch.cuda.current_stream() is by default)
  return 0;
}

int i3rc_hip_use_own_stream(i3rc_hip_integrator *h) {
  if (!h) return 1;
  h->stream = h->ownStream;
  return 0;
}

int i3rc_hip_zero_tallies(i3rc_hip_integrator *h) {
  if (!h) return 1;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemsetAsync(h->tally, 0, (size_t)h->layout.total * sizeof(double), h->stream));
  return 0;
}

/* Tunables for experiments (not part of the reference API): event-phase ballot threshold, blocks per CU. */
int i3rc_hip_set_tuning(i3rc_hip_integrator *h, int evThreshold, int blocksPerCU) {
  if (!h) return 1;
  drop_lookahead(h);
  if (evThreshold >= 0 && evThreshold <= 64) h->evThreshold = evThreshold;   // 0 = default
  if (blocksPerCU >= 0 && blocksPerCU <= 8) h->blocksPerCU = blocksPerCU;
  return 0;
}

int i3rc_hip_set_light_threshold(i3rc_hip_integrator *h, int lanes) {
  if (!h) return 1;
  drop_lookahead(h);
  if (lanes < 0 || lanes > 64) return h->fail("i3rc_hip_set_light_threshold: need 1..64 lanes (0 = default)");
  h->lightThreshold = lanes;
  return 0;
}

int i3rc_hip_set_launch_limit(i3rc_hip_integrator *h, int64_t photons) {
  if (!h) return 1;
  drop_lookahead(h);
  if (photons < 0) return h->fail("i3rc_hip_set_launch_limit: negative limit");
  h->launchLimit = photons;
  return 0;
}

int i3rc_hip_select_kernel(i3rc_hip_integrator *h, int variant) {
  if (!h) return 1;
  drop_lookahead(h);
  if (variant < I3RC_KERNEL_AUTO || variant > I3RC_KERNEL_RING) return h->fail("i3rc_hip_select_kernel: unknown variant");
  h->kernelVariant = variant;
  return 0;
}

int i3rc_hip_select_grid_place(i3rc_hip_integrator *h, int place) {
  if (!h) return 1;
  drop_lookahead(h);
  if (place < I3RC_GRID_AUTO || place > I3RC_GRID_COLUMNS) return h->fail("i3rc_hip_select_grid_place: unknown place");
  if (place == I3RC_GRID_COLUMNS && !h->dColRec.p)
    return h->fail("i3rc_hip_select_grid_place: the field has no column records (some column holds more than one run of one value)");
  if (place == I3RC_GRID_BRICKS && h->nz > 65534) return h->fail("i3rc_hip_select_grid_place: more than 65534 layers keep the linear field");
  h->gridPlace = place;
  return 0;
}

int i3rc_hip_has_column_records(const i3rc_hip_integrator *h) { return h && h->dColRec.p ? 1 : 0; }

/* Column records (DevProblem::colRec) of a field [nz][ny][nx]: possible when the cells with extinction of every column are ONE run
 * of layers that hold ONE value -- compared bit by bit; "no extinction" is the bit pattern of +0, which is what a record gives
 * outside its run (a -0, a NaN or a negative value is "something there" and must be the run's value like any other).  Host code
 * only.  records (may be NULL): [ny * nx][2] words -- the value's bits; first layer (1-based) | (run length - 1) << 16; a clear
 * column is the value 0 in layer 1.  Returns 1 when the field has the form, 0 when it has not (or has more than 65534 layers). */
int i3rc_hip_column_records(int nx, int ny, int nz, const float *totalExt, uint32_t *records) {
  if (nx < 1 || ny < 1 || nz < 1 || !totalExt || nz > 65534) return 0;
  const size_t ncol = (size_t)nx * ny;
  for (size_t c = 0; c < ncol; ++c) {
    uint32_t val = 0u; int first = 0, last = 0;   // 1-based layers of the run
    for (int k = 0; k < nz; ++k) {
      uint32_t bits; std::memcpy(&bits, &totalExt[(size_t)k * ncol + c], sizeof(bits));
      if (bits == 0u) continue;
      if (first == 0) { val = bits; first = last = k + 1; }
      else if (bits == val && last == k) last = k + 1;
      else return 0;
    }
    if (records) {
      records[2 * c] = val;
      records[2 * c + 1] = first == 0 ? 1u : ((uint32_t)first | ((uint32_t)(last - first) << 16));
    }
  }
  return 1;
}

/* Column records OVER A BASE PROFILE (DevProblem::colBase): the field is base(z) + (z within the column's run ? value(x, y) : 0) in
 * float32 arithmetic -- what a cloud scene of one run of one value per column and a horizontally uniform second component add up to.
 * base(z) is the smallest value of the layer (some column must be outside its run there); a column's run is where it differs from
 * the base, and its value is looked for among the float32 numbers around (first differing value - base): the one whose float32 sum
 * with the base gives the field's value in EVERY layer of the run, bit by bit.  Host code only.  Returns 1 / 0; records as
 * i3rc_hip_column_records, base[nz]. */
int i3rc_hip_column_records_base(int nx, int ny, int nz, const float *totalExt, uint32_t *records, float *base) {
  if (nx < 1 || ny < 1 || nz < 1 || !totalExt || !base || nz > 65534) return 0;
  const size_t ncol = (size_t)nx * ny;
  auto bits_of = [](float v) { uint32_t b; std::memcpy(&b, &v, sizeof(b)); return b; };
  for (int k = 0; k < nz; ++k) {
    float b = totalExt[(size_t)k * ncol];
    for (size_t c = 1; c < ncol; ++c) b = std::min(b, totalExt[(size_t)k * ncol + c]);
    if (!(b >= 0.0f) || !std::isfinite(b)) return 0;
    base[k] = b;
  }
  for (size_t c = 0; c < ncol; ++c) {
    int first = 0, last = 0;
    for (int k = 0; k < nz; ++k) {
      if (bits_of(totalExt[(size_t)k * ncol + c]) == bits_of(base[k])) continue;
      if (first == 0) first = last = k + 1;
      else if (last == k) last = k + 1;
      else return 0;                       // a second run
    }
    uint32_t val = 0u;
    if (first != 0) {
      const float guess = totalExt[(size_t)(first - 1) * ncol + c] - base[first - 1];
      bool found = false;
      for (int step = 0; step <= 8 && !found; ++step) {   // the candidates: guess + j float32 steps, j = 0, +1, -1, +2, -2, ...
        const int j = (step + 1) / 2 * (step % 2 ? 1 : -1);
        float v = guess;
        for (int t = 0; t < std::abs(j); ++t) v = std::nextafter(v, j > 0 ? INFINITY : -INFINITY);
        if (!(v > 0.0f)) continue;
        bool ok = true;
        for (int k = first - 1; k < last && ok; ++k) {
          volatile float sum = base[k] + v;   // (one float32 addition, as the host that summed the components made it)
          ok = bits_of(sum) == bits_of(totalExt[(size_t)k * ncol + c]);
        }
        if (ok) { val = bits_of(v); found = true; }
      }
      if (!found) return 0;
    }
    if (records) {
      records[2 * c] = val;
      records[2 * c + 1] = first == 0 ? 1u : ((uint32_t)first | ((uint32_t)(last - first) << 16));
    }
  }
  return 1;
}

int i3rc_hip_set_lds_tallies(i3rc_hip_integrator *h, int on) {
  if (!h) return 1;
  drop_lookahead(h);
  h->ldsTalliesOn = on != 0;
  return 0;
}

int i3rc_hip_lds_plan(const int32_t *q, int32_t *out) {
  if (!q || !out) return 1;
  DevProblem P;
  std::memset(&P, 0, sizeof(P));
  P.nx = q[0]; P.ny = q[1]; P.nz = q[2]; P.ncomp = q[3]; P.nDir = q[4]; P.ldsTallies = q[5]; P.ldsIntensity = q[6];
  P.rayQueueCap = q[7]; P.clearNx = q[8]; P.clearShift = q[9]; P.ldsVolume = q[16];
  const LdsPlan lp = lds_plan(P, q[10] != 0, q[11] != 0, q[12], q[13] != 0, q[14], q[15]);
  const int v[12] = {lp.xE, lp.yE, lp.zE, lp.tallies, lp.dirCos, lp.dirTab, lp.queue, lp.tInt, lp.ext, lp.cosTab, lp.end, lp.tVol};
  for (int k = 0; k < 12; ++k) out[k] = v[k];
  return 0;
}

int i3rc_hip_set_batch_fusion(i3rc_hip_integrator *h, int mode) {
  if (!h) return 1;
  drop_lookahead(h);
  if (mode < -1 || mode > 1) return h->fail("i3rc_hip_set_batch_fusion: mode is -1 (automatic), 0 (never) or 1 (whenever possible)");
  h->fusion = mode;
  h->fusedAheadFailed = false;
  return 0;
}

/* Older name of i3rc_hip_select_kernel(h, on ? I3RC_KERNEL_GENERAL : I3RC_KERNEL_AUTO). */
int i3rc_hip_force_general_kernel(i3rc_hip_integrator *h, int on) {
  return i3rc_hip_select_kernel(h, on ? I3RC_KERNEL_GENERAL : I3RC_KERNEL_AUTO);
}

}  // extern "C"

namespace {

struct LaunchPlan {
  DevProblem P;
  size_t ldsBytes;
  bool intensity;
  int place;     // GridPlace: where the kernels read the extinction field (make_problem)
};

constexpr size_t kLdsBudget = 64 * 1024;  // per workgroup: leaves room for >= 2 workgroups per CU
// ... which what a launch MUST have in LDS -- the edge vectors, the directions, the ray queues -- may exceed, up to the 160 KB of a
// compute unit (less the kernels' few static words): a 2-D domain of 20 000 columns runs with one workgroup per CU, slowly,
// instead of being refused
constexpr size_t kLdsHard = 158 * 1024;

// Which kernel runs a launch (see photon_kernel): the common problem class -- regular grid,
// ray tracing, one component, no BRDF grid, Directional source -- has specialised kernels.
// A surface description with a single cell (new_SurfaceDescription((/ albedo /)), the form BASELINE.json's Landsat
// radiance case uses) reflects like surfaceAlbedo: computeSurfaceReflectance returns its one parameter wherever the
// photon lands (Code/surfaceProperties.f95:121-162), and the weight is multiplied by the same float.
bool uniform_surface(const i3rc_hip_integrator *h) { return h->params.useSurfaceBDRF && h->nxs == 1 && h->nys == 1; }

// ray tracing asked for, or max cross-section on an optically empty domain (see make_problem)
bool traced(const i3rc_hip_integrator *h) {
  const float width = std::min(h->xE.back() - h->xE.front(), h->yE.back() - h->yE.front());
  return h->params.useRayTracing || !(h->maxExt * width > 1e-5f);
}

bool common_class(const i3rc_hip_integrator *h, int srcKind) {
  const bool gridSurface = h->params.useSurfaceBDRF && !uniform_surface(h);
  return h->xyRegular && traced(h) && !gridSurface && h->ncomp == 1 && srcKind == 0;
}
// ... and the same class widened: several components (photon_kernel, MULTI; round 5).  I3RC_MULTI=0 leaves such problems to the general kernels.
// (... and, since the kernels that run it keep those two paths behind run-time switches, with an irregular x / y grid or a gridded surface)
bool multi_class(const i3rc_hip_integrator *h, int srcKind) {
  static const bool on = !(std::getenv("I3RC_MULTI") && std::atoi(std::getenv("I3RC_MULTI")) == 0);
  return on && traced(h) && srcKind == 0;
}

// One radiance direction (nadir views: BASELINE.json's radar case): the radiance kernels without an event ring (photon_kernel,
// DIRECT).  I3RC_DIRECT=0 keeps the ring for them too.
#ifdef I3RC_NESTED_BUILD   /* measurement build: radiance problems run the general kernels with the nested local estimate (kernels.hpp) */
constexpr bool kNestedBuild = true;
#else
constexpr bool kNestedBuild = false;
#endif
bool direct_rays(const i3rc_hip_integrator *h) {
  static const bool on = !(std::getenv("I3RC_DIRECT") && std::atoi(std::getenv("I3RC_DIRECT")) == 0);
  return on && h->nDir == 1 && h->kernelVariant != I3RC_KERNEL_RING && !kNestedBuild;
}

size_t ncell_bytes(const i3rc_hip_integrator *h) { return sizeof(float) * (size_t)h->nx * h->ny * h->nz; }

int make_problem(i3rc_hip_integrator *h, LaunchPlan &plan, bool fused = false, bool replay = false) {
  DevProblem &P = plan.P;
  std::memset(&P, 0, sizeof(P));
  for (int c = 0; c < h->ncomp; ++c)
    if (!h->comp[c].inv) return h->fail("computeRadiativeTransfer: problem not completely specified (inverse phase function table missing).");
  if (h->nDir > 0)
    for (int c = 0; c < h->ncomp; ++c)
      if (!h->comp[c].fwd) return h->fail("computeRadiativeTransfer: problem not completely specified (forward phase function table missing).");
  P.nx = h->nx; P.ny = h->ny; P.nz = h->nz; P.ncomp = h->ncomp;
  P.xyRegular = h->xyRegular; P.zRegular = h->zRegular;
  P.x0 = h->xE.front(); P.xMax = h->xE.back();
  P.y0 = h->yE.front(); P.yMax = h->yE.back();
  P.z0 = h->zE.front(); P.zMax = h->zE.back();
  P.deltaX = h->xE[1] - h->xE[0]; P.deltaY = h->yE[1] - h->yE[0]; P.deltaZ = h->zE[1] - h->zE[0];
  P.xE = (const float *)h->dxE.p; P.yE = (const float *)h->dyE.p; P.zE = (const float *)h->dzE.p;
  // bricks pay off once the field no longer fits in one XCD's 4 MB of L2
  // (the clear-air map of a bricked field holds layer numbers in 16 bits: domains of more layers than that keep the linear field)
  // Column records where the field has them (and does not fit in LDS, below): the whole field in 8 bytes per column.  Measured:
  // I3RC_COLUMNS=0 switches them off for the process.
  static const bool columnsOn = !(std::getenv("I3RC_COLUMNS") && std::atoi(std::getenv("I3RC_COLUMNS")) == 0);
  // (records over a base profile -- GRID_COLBASE -- are read by the kernels of domains with several components, the general and the
  // several-components ones: the one-component specialisations and the replay build are not instantiated for them)
  const bool baseForm = h->dColBase.p != nullptr;
  const bool baseOk = h->ncomp > 1 && !replay;
  const bool columns = (h->gridPlace == I3RC_GRID_COLUMNS || (h->gridPlace == I3RC_GRID_AUTO && columnsOn && h->dColRec.p != nullptr)) && (!baseForm || baseOk);
  P.colRec = columns ? (const uint2 *)h->dColRec.p : nullptr;
  P.colBase = columns && baseForm ? (const float *)h->dColBase.p : nullptr;
  const bool bricks = h->gridPlace == I3RC_GRID_BRICKS || (h->gridPlace == I3RC_GRID_AUTO && !columns && ncell_bytes(h) > ((size_t)4 << 20) && h->nz <= 65534);
  P.extBrick = bricks ? (const float *)h->dExtBrick.p : nullptr;
  P.bsx = h->bsx; P.bsy = h->bsy; P.bsz = h->bsz; P.nbx = h->nbx; P.nbxy = h->nbx * h->nby;
  P.clearMap = (const uint32_t *)h->dClearMap.p; P.clearShift = h->clearShift; P.clearNx = h->clearNx;
  P.totalExt = (const float *)h->dExt.p; P.cumExt = (const float *)h->dCum.p; P.ssa = (const float *)h->dSsa.p;
  P.pfIndex = (const int32_t *)h->dPf.p;
  P.cellRec = (const uint4 *)h->dCellRec.p;
  if (h->compDirty) {
    // the device copy of the table descriptors follows the host copy when a table was (re)set: a blocking copy after
    // the stream has drained (launches in flight read the old descriptors), not an asynchronous copy from the
    // pageable handle per launch
    if (hipStreamSynchronize(h->stream) != hipSuccess ||
        hipMemcpy(h->dComp.p, h->comp.data(), sizeof(CompTables) * h->ncomp, hipMemcpyHostToDevice) != hipSuccess)
      return h->fail("copying the component table descriptors failed");
    h->compDirty = false;
  }
  P.comp = (const CompTables *)h->dComp.p;
  P.comp0 = h->comp[0];
  P.albedo = h->params.surfaceAlbedo; P.useBDRF = h->params.useSurfaceBDRF;
  if (uniform_surface(h)) { P.albedo = h->brdf0; P.useBDRF = 0; }
  P.nxs = h->nxs; P.nys = h->nys;
  P.xsE = (const float *)h->dXs.p; P.ysE = (const float *)h->dYs.p; P.brdf = (const float *)h->dBrdf.p;
  if (P.useBDRF && !P.brdf) return h->fail("computeRadiativeTransfer: surfaceBDRF requested but no surface description set");
  // Max cross-section divides the optical depth by the largest extinction of the domain (:494-496): with no extinction
  // anywhere that is a step of infinite length and the reference's makePeriodic never returns -- nor does it once the
  // step exceeds 2^24 domain widths, where subtracting a width no longer changes a float32.  A photon in such an
  // (optically empty: width * maxExtinction <= 1e-5) domain flies straight to the boundary, which is what ray
  // tracing gives: such a domain is traced.
  P.useRayTracing = traced(h) ? 1 : 0; P.useRR = h->params.useRussianRoulette;
  P.nDir = h->nDir; P.useHybrid = h->params.useHybridPhaseFunsForIntenCalcs;
  P.numOrdersOrig = h->params.numOrdersOrigPhaseFunIntenCalcs; P.useRRI = h->params.useRussianRouletteForIntensity;
  P.limitContrib = h->params.limitIntensityContributions; P.zetaMin = h->params.zetaMin;
  P.maxContrib = h->params.maxIntensityContribution; P.maxExt = h->maxExt;
  P.dirCos = (const float *)h->dDir.p;
  P.uniformSsa = h->uniformSsa; P.uniformPf = h->uniformPf;
  if (h->layout.total >= ((int64_t)1 << 31)) return h->fail("tally buffer too large (2^31 elements or more)");
  P.tally = h->tally;
  P.oUp = (int)h->layout.fluxUp; P.oDown = (int)h->layout.fluxDown; P.oAbs = (int)h->layout.fluxAbsorbed;
  P.oVol = (int)h->layout.volumeAbsorption; P.oInt = (int)h->layout.intensityByComponent;
  P.oExc = (int)h->layout.intensityExcess; P.oCnt = (int)h->layout.counters;
  const size_t ncol = (size_t)h->nx * h->ny, ncell = ncol * h->nz;
  size_t lds = sizeof(float) * ((h->nx + 1) + (h->ny + 1) + (h->nz + 1) + 3 * (size_t)h->nDir);
  // radiance runs: every wave's ring of local-estimate events (one record serves the nDir rays of an event) and its
  // buffer of ready-made rays (photon_kernel, ray mode)
  // (one direction: no ring, a ready store of two wavefronts -- photon_kernel, DIRECT)
  P.rayQueueCap = h->nDir > 0 && !direct_rays(h) ? 64 : 0;   // (an event phase pushes at most 64 records; the rays go on to the ready buffer)
  if (h->nDir > 0) lds += sizeof(float) * 4 * (kRecWords * (size_t)P.rayQueueCap + kReadyWords * (size_t)(direct_rays(h) ? kDirectReady : kReadyRays));
  if (h->nDir > 0) lds += sizeof(float) * (16 * (size_t)h->nDir + 3);   // per direction: what a ray derives from it (Lds::dirTab, 16-byte aligned)
  // (`lds` steers the decisions below and is an upper bound; what a launch allocates is lds_plan's own end: lds_bytes)
  if (h->nDir > 0)
    for (int c = 0; c < h->ncomp; ++c)
      if (h->maxPfIndex[c] >= 65536) return h->fail("radiance runs take at most 65535 phase-function table entries per component");
  if (lds > kLdsHard) return h->fail("domain edge vectors do not fit in LDS (nx + ny + nz beyond about 39 000)");
  const size_t budget = kLdsBudget;
  P.ldsTallies = 0;
  // (a fused multi-batch launch tallies per batch, straight into global memory: no partial sums in LDS)
  // (float64 partial sums: tracer.hpp, tally_t; + 4: their 8-byte alignment.  I3RC_LDS_TALLIES=0 / i3rc_hip_set_lds_tallies(h, 0): every
  // tally straight to the float64 buffer in global memory -- a measurement knob, and one more order of the same float64 additions)
  static const bool ldsTalliesEnv = !(std::getenv("I3RC_LDS_TALLIES") && std::atoi(std::getenv("I3RC_LDS_TALLIES")) == 0);
  const bool privatise = !fused && ldsTalliesEnv && h->ldsTalliesOn;
  if (privatise && lds + 2 * ncol * sizeof(tally_t) + 4 <= kLdsBudget / 2) { P.ldsTallies = 1; lds += 2 * ncol * sizeof(tally_t) + 4; }
  // (an absorbing domain of few cells -- the step cloud's 512 or 1024 --: its volume-absorption tallies, which every scattering adds to)
  P.ldsVolume = 0;
  if (privatise && h->absorbing && lds + ncell * sizeof(tally_t) + 4 <= kLdsBudget / 2) { P.ldsVolume = 1; lds += ncell * sizeof(tally_t) + 4; }
  P.ldsIntensity = 0;
  {
    const size_t nInt = (size_t)(h->ncomp + 1) * h->nDir * ncol * sizeof(tally_t) + 4;
    if (privatise && h->nDir > 0 && nInt <= 16 * 1024 && lds + nInt <= kLdsBudget) { P.ldsIntensity = 1; lds += nInt; }
  }
  P.ldsGrid = 0;
  if (h->gridPlace == I3RC_GRID_AUTO && lds + ncell * sizeof(float) <= budget) { P.ldsGrid = 1; lds += ncell * sizeof(float); P.colRec = nullptr; P.colBase = nullptr; P.extBrick = nullptr; }   // (never when the edges alone are beyond the budget)
  else if (P.extBrick && h->nDir == 0) lds += sizeof(uint32_t) * (size_t)h->clearWords;          // bricked field, flux kernels: its clear-air map
  if (P.colBase) lds += sizeof(float) * (size_t)h->nz;                                            // column records over a base profile: the profile
  plan.ldsBytes = (lds + 15) & ~(size_t)15;
  plan.intensity = h->nDir > 0;
  plan.place = P.ldsGrid ? GRID_LDS : (P.colRec ? (P.colBase ? GRID_COLBASE : GRID_COLUMNS) : (P.extBrick ? GRID_BRICKS : GRID_GLOBAL));
  return 0;
}

// fluxAbsorbed(ix, iy) of a tally block = the sum of its column's volumeAbsorption (:644-647 add the same increment to both; the kernels
// tally the cell only: Tally::absorbed).  An assignment: a block that several launches add to is summed again after each of them.
__global__ void __launch_bounds__(256) absorbed_columns_kernel(double *blocks, long long stride, int oAbs, int oVol, int ncol, int nz) {
  const int col = (int)(blockIdx.x * 256 + threadIdx.x);
  if (col >= ncol) return;
  double *const blk = blocks + (size_t)blockIdx.y * stride;
  double s = 0.0;
  for (int k = 0; k < nz; ++k) s += blk[oVol + (size_t)k * ncol + col];
  blk[oAbs + col] = s;
}
int absorbed_columns(i3rc_hip_integrator *h, hipStream_t stream, double *blocks, int nBlocks, long long stride) {
  if (!h->absorbing || nBlocks < 1) return 0;   // (nothing absorbs: both tallies stay zero)
  const int ncol = h->nx * h->ny;
  for (int first = 0; first < nBlocks; first += 65535) {
    const int n = std::min(65535, nBlocks - first);
    hipLaunchKernelGGL(absorbed_columns_kernel, dim3((unsigned)((ncol + 255) / 256), (unsigned)n), dim3(256), 0, stream, blocks + (size_t)first * stride,
                       stride, (int)h->layout.fluxAbsorbed, (int)h->layout.volumeAbsorption, ncol, h->nz);
    HIPCHK(h, hipGetLastError());
  }
  return 0;
}

// Dynamic LDS of one launch: the end of the kernel's own carve-up (lds_plan, tracer.hpp -- the function photon_kernel sets its
// pointers from), for the instantiation that is about to run.
template <class Rng>
size_t lds_bytes(const i3rc_hip_integrator *h, const LaunchPlan &plan, bool tableInLds) {
  const LdsPlan lp = lds_plan(plan.P, plan.intensity && !Rng::kReplay, direct_rays(h), plan.place, plan.intensity, tableInLds ? 16 : 4,
                              tableInLds ? plan.P.comp0.nInv : 0);
  return (sizeof(float) * (size_t)lp.end + 15) & ~(size_t)15;
}

int upload_source(i3rc_hip_integrator *h, const i3rc_source *src, int64_t n, RunArgs &A) {
  A.srcKind = src->kind;
  if (src->kind == 0) {
    if (std::fabs(src->solarMu) > 1.f || std::fabs(src->solarMu) <= FLT_MIN) return h->fail("setIllumination: solarMu out of bounds");
    if (src->solarAzimuth < 0.f || src->solarAzimuth > 360.f) return h->fail("setIllumination: solarAzimuth out of bounds");
    A.solarMu = -std::fabs(src->solarMu);                      // Code/monteCarloIllumination.f95:98
    A.solarPhi = src->solarAzimuth * std::acos(-1.0f) / 180.f; // :99
    // makeDirectionCosines (:2041-2059) once on the host: the same libm float32 calls the reference makes
    const float sinTheta = std::sqrt(1.0f - A.solarMu * A.solarMu);
    A.solarDx = sinTheta * std::cos(A.solarPhi);
    A.solarDy = sinTheta * std::sin(A.solarPhi);
    A.solarDz = A.solarMu;
    return 0;
  }
  if (src->kind != 1) return h->fail("unknown photon source kind");
  const float *arrs[5] = {src->x, src->y, src->z, src->mu, src->phi};
  const float **dst[5] = {&A.sx, &A.sy, &A.sz, &A.smu, &A.sphi};
  for (int k = 0; k < 5; ++k) if (!arrs[k]) return h->fail("explicit photon stream: null array");
  // what the reference's photon-stream constructors check (Code/monteCarloIllumination.f95:78-83, :124, :204, :369-371):
  // relative positions within [0, 1], |mu| in (tiny, 1] -- a horizontal or NaN direction would never leave the domain
  for (int64_t i = 0; i < n; ++i) {
    const float mu = src->mu[i];
    if (!(std::fabs(mu) <= 1.f) || !(std::fabs(mu) > FLT_MIN)) return h->fail("setIllumination: solarMu out of bounds");
    if (!(src->x[i] >= 0.f && src->x[i] <= 1.f) || !(src->y[i] >= 0.f && src->y[i] <= 1.f) || !(src->z[i] >= 0.f && src->z[i] <= 1.f))
      return h->fail("setIllumination: photon position out of bounds (relative positions lie in [0, 1])");
    if (!std::isfinite(src->phi[i])) return h->fail("setIllumination: solarAzimuth out of bounds");
  }
  for (int k = 0; k < 5; ++k) {
    HIPCHK(h, h->srcBuf[k].upload(arrs[k], sizeof(float) * (size_t)n));
    *dst[k] = (const float *)h->srcBuf[k].p;
  }
  return 0;
}

template <class Rng, bool INTENSITY, bool GENERAL, bool DIRECT, bool MULTI>
constexpr void (*colbase_kernel())(DevProblem, RunArgs, int, int) {
  if constexpr (Rng::kReplay) return nullptr;
  else return photon_kernel<Rng, INTENSITY, GENERAL, GRID_COLBASE, false, DIRECT, MULTI>;
}

template <class Rng>
int launch(i3rc_hip_integrator *h, const LaunchPlan &plan, const RunArgs &A, bool timeIt) {
  // fast specialisations when the problem is in the common class (see photon_kernel), else the general kernel
  const bool simple = !Rng::kReplay && common_class(h, A.srcKind) && h->kernelVariant != I3RC_KERNEL_GENERAL && !(kNestedBuild && plan.intensity);
  // the specialised kernels exist once per place of the extinction grid (LDS / global / global in bricks)
  using Kernel = void (*)(DevProblem, RunArgs, int, int);
  const int place = plan.place;
  // (GRID_COLBASE -- column records over a base profile -- exists for the kernels that run domains of several components: the general
  // ones and the several-components ones; make_problem never plans it for anything else)
  static const Kernel general[2][5] = {
      {photon_kernel<Rng, false, true, GRID_LDS>, photon_kernel<Rng, false, true, GRID_GLOBAL>, photon_kernel<Rng, false, true, GRID_BRICKS>, photon_kernel<Rng, false, true, GRID_COLUMNS>, colbase_kernel<Rng, false, true, false, false>()},
      {photon_kernel<Rng, true, true, GRID_LDS>, photon_kernel<Rng, true, true, GRID_GLOBAL>, photon_kernel<Rng, true, true, GRID_BRICKS>, photon_kernel<Rng, true, true, GRID_COLUMNS>, colbase_kernel<Rng, true, true, false, false>()}};
  Kernel kern = general[plan.intensity ? 1 : 0][place];
  if constexpr (!Rng::kReplay) {   // (the replay build always runs the general kernel)
    static const Kernel special[2][5] = {
        {photon_kernel<Rng, false, false, GRID_LDS>, photon_kernel<Rng, false, false, GRID_GLOBAL>, photon_kernel<Rng, false, false, GRID_BRICKS>, photon_kernel<Rng, false, false, GRID_COLUMNS>, nullptr},
        {photon_kernel<Rng, true, false, GRID_LDS>, photon_kernel<Rng, true, false, GRID_GLOBAL>, photon_kernel<Rng, true, false, GRID_BRICKS>, photon_kernel<Rng, true, false, GRID_COLUMNS>, nullptr}};
    if (simple) kern = special[plan.intensity ? 1 : 0][place];
    // several components, otherwise the common class: RADIANCE problems run photon_kernel<..., MULTI> (+20 % on the Landsat scene + gas
    // with seven directions against the general radiance kernels' 166 registers and three waves per SIMD).  Flux problems stay with
    // the general flux kernel: its several-components specialisation was built and measured -- 5.78 against 5.71e8 photons/s on
    // Landsat-119 + gas, 9.13 against 9.08e8 on Landsat-36 + gas: the voxel steps are the same code, and a flux event's few extra
    // reads do not show (profiles/r05_ab_experiments.txt) -- and is not in the tree.
    const bool multi = !simple && plan.intensity && multi_class(h, A.srcKind) && h->kernelVariant != I3RC_KERNEL_GENERAL && !kNestedBuild;
    if (multi) {
      static const Kernel several[2][5] = {
          {photon_kernel<Rng, true, false, GRID_LDS, false, false, true>, photon_kernel<Rng, true, false, GRID_GLOBAL, false, false, true>, photon_kernel<Rng, true, false, GRID_BRICKS, false, false, true>, photon_kernel<Rng, true, false, GRID_COLUMNS, false, false, true>, photon_kernel<Rng, true, false, GRID_COLBASE, false, false, true>},
          {photon_kernel<Rng, true, false, GRID_LDS, false, true, true>, photon_kernel<Rng, true, false, GRID_GLOBAL, false, true, true>, photon_kernel<Rng, true, false, GRID_BRICKS, false, true, true>, photon_kernel<Rng, true, false, GRID_COLUMNS, false, true, true>, photon_kernel<Rng, true, false, GRID_COLBASE, false, true, true>}};
      kern = several[direct_rays(h) ? 1 : 0][place];
    } else
    if (plan.intensity && direct_rays(h)) {   // (the replay build keeps the nested local estimate: no queue at all)
      static const Kernel direct[2][5] = {
          {photon_kernel<Rng, true, true, GRID_LDS, false, true>, photon_kernel<Rng, true, true, GRID_GLOBAL, false, true>, photon_kernel<Rng, true, true, GRID_BRICKS, false, true>, photon_kernel<Rng, true, true, GRID_COLUMNS, false, true>, colbase_kernel<Rng, true, true, true, false>()},
          {photon_kernel<Rng, true, false, GRID_LDS, false, true>, photon_kernel<Rng, true, false, GRID_GLOBAL, false, true>, photon_kernel<Rng, true, false, GRID_BRICKS, false, true>, photon_kernel<Rng, true, false, GRID_COLUMNS, false, true>, nullptr}};
      kern = direct[simple ? 1 : 0][place];
    }
  }
  // Flux problems of the common class with ONE phase-function entry keep the inverse table's cosines (40 KB) in LDS, in
  // workgroups of 1024 threads, two per compute unit (photon_kernel, TBL): the two dependent table reads of a scattering come
  // from LDS instead of L2 -- or, where the extinction field fills the L2 (Landsat-36: 2.4 MB of 4), instead of the fabric.
  // Step cloud 29.75 -> 29.29 ms per 1e8 photons (+1.6 %), radar 640 +2 %, Landsat-36 87.0 -> 71.0 ms (+22 %).
  // I3RC_TABLE_LDS=0 switches it off.
  int threads = 256;
  size_t ldsBytes = lds_bytes<Rng>(h, plan, false);
  if constexpr (!Rng::kReplay) {
    static const bool tblOn = !(std::getenv("I3RC_TABLE_LDS") && std::atoi(std::getenv("I3RC_TABLE_LDS")) == 0);
    // (grid places as a bit mask: LDS and global memory.  Bricked fields: Landsat-119 -2.5 %, the scene tiled 2 x 2 +10 %: left out)
    static const int tblPlaces = std::getenv("I3RC_TABLE_LDS_PLACES") ? std::atoi(std::getenv("I3RC_TABLE_LDS_PLACES")) : 11;
    if (tblOn && simple && !plan.intensity && place != GRID_COLBASE && ((tblPlaces >> place) & 1) && (plan.P.uniformPf >= 1 || h->nInvEntries[0] == 1) && h->kernelVariant == I3RC_KERNEL_AUTO &&
        plan.ldsBytes + sizeof(float) * (size_t)plan.P.comp0.nInv <= 79 * 1024) {
      static const Kernel tbl[4] = {photon_kernel<Rng, false, false, GRID_LDS, true>, photon_kernel<Rng, false, false, GRID_GLOBAL, true>,
                                    photon_kernel<Rng, false, false, GRID_BRICKS, true>, photon_kernel<Rng, false, false, GRID_COLUMNS, true>};
      kern = tbl[place];
      threads = 1024;
      ldsBytes = lds_bytes<Rng>(h, plan, true);
    }
  }
  if (ldsBytes > 160 * 1024 - 256) return h->fail("the launch needs more LDS than a compute unit has");
  if (!kern) return h->fail("internal: no kernel for this problem at this place of the extinction field");
  const void *fn = (const void *)kern;
  {
    static const char *const placeName[5] = {"GRID_LDS", "GRID_GLOBAL", "GRID_BRICKS", "GRID_COLUMNS", "GRID_COLBASE"};
    static thread_local char name[96];
    std::snprintf(name, sizeof(name), "photon_kernel<%s, %s, %s, %s>", Rng::kReplay ? "ReplayStream" : "PhiloxStream",
                       plan.intensity ? "true" : "false", (simple ? "false" : "true"), placeName[place]);
    h->lastKernelName = name;
    if (threads == 1024) { std::snprintf(name, sizeof(name), "photon_kernel<PhiloxStream, false, false, %s, table in LDS>", placeName[place]); h->lastKernelName = name; }
    if (!Rng::kReplay && plan.intensity && direct_rays(h)) {
      std::snprintf(name, sizeof(name), "photon_kernel<PhiloxStream, true, %s, %s, one direction>", (simple ? "false" : "true"), placeName[place]);
      h->lastKernelName = name;
    }
    if (!Rng::kReplay && !simple && plan.intensity && multi_class(h, A.srcKind) && h->kernelVariant != I3RC_KERNEL_GENERAL && !kNestedBuild) {
      std::snprintf(name, sizeof(name), "photon_kernel<PhiloxStream, %s, false, %s%s, wide>", plan.intensity ? "true" : "false", placeName[place],
                    plan.intensity && direct_rays(h) ? ", one direction" : "");
      h->lastKernelName = name;
    }
  }
  int perCU = h->blocksPerCU;
  if (perCU <= 0) {
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, threads, ldsBytes) != hipSuccess || occ < 1) occ = 2;
    perCU = std::min(occ, 8);
    // a field beyond an XCD's L2 (bricks): every wave in flight widens the part of it that is in use; measured on the
    // 7.8 MB Landsat-119 field (tools/blocks_sweep.py): 4-5 workgroups per CU 6.17e8 photons/s, 6-8 5.83e8.  (The radiance
    // kernels have the registers for five at most; fields within L2 gain up to 7.)
    // With the XCD-aware photon order: Landsat-119 5 ... 8 alike (6.5e8); the scene tiled 2 x 2 (31 MB): 4 workgroups 5.41e8,
    // 5 5.02e8, 6-8 4.6e8.
    if (place == GRID_BRICKS) perCU = std::min(perCU, ncell_bytes(h) > ((size_t)16 << 20) ? 4 : 5);
  }
  if (ldsBytes > 48 * 1024)
    HIPCHK(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
  long long blocks = (long long)h->numCU * perCU;
  const long long need = (A.nPhotons + threads - 1) / threads;
  if (blocks > need) blocks = std::max(1ll, need);
  RunArgs B = A;   // photon indices are handed to waves in chunks (one returning atomic per chunk)
  // XCD-aware photon order.  A field beyond an XCD's L2 (bricks) is shared by eight L2s that each see all of it: when
  // the photons a wave traces start in "its" eighth of the domain -- workgroups are dealt round-robin over the XCDs, the
  // kernel reads its XCC id -- an XCD's L2 has an eighth of the field (and its surroundings) to hold.  The launch's
  // photons are sorted by start slab first (their start position is their own first Philox block): two passes over the
  // photon numbers, about 1 % of the launch.  Radiance runs: local-estimate rays cross much of the domain, but from one
  // tile they keep to a tube per direction -- nothing gained on the 7.8 MB Landsat field (9.49 against 9.45e7 photons/s:
  // left in index order), +20 % on a 31 MB field (6.9 -> 8.3e7 with 7 directions).
  // Measured ceiling (tools/locality_experiment.py): +9 ... 14 % on the 7.8 MB Landsat field, +36 % on a 62 MB field.
  // I3RC_SLABS=0 switches it off.
  static const bool slabsOn = !(std::getenv("I3RC_SLABS") && std::atoi(std::getenv("I3RC_SLABS")) == 0);
  B.slabIds = nullptr; B.slabMeta = nullptr;
  if constexpr (!Rng::kReplay) {
    if (slabsOn && place == GRID_BRICKS && (!plan.intensity || ncell_bytes(h) > ((size_t)16 << 20)) && A.srcKind == 0 && A.nPhotons >= 1024 && A.nPhotons < ((long long)1 << 32)) {
      auto &sb = h->slabBufs[h->stream];
      if (sb.ids.bytes < (size_t)A.nPhotons * sizeof(uint32_t)) HIPCHK(h, sb.ids.alloc((size_t)A.nPhotons * sizeof(uint32_t)));
      if (!sb.meta.p || sb.meta.bytes != sizeof(SlabMeta)) HIPCHK(h, sb.meta.alloc(sizeof(SlabMeta)));
      const size_t tableBytes = sizeof(unsigned) * 8 * kSlabSortBlocks;
      if (sb.blockCounts.bytes != tableBytes) { HIPCHK(h, sb.blockCounts.alloc(tableBytes)); HIPCHK(h, sb.blockBase.alloc(tableBytes)); }
      const long long span = ((A.nPhotons + kSlabSortBlocks - 1) / kSlabSortBlocks + 255) / 256 * 256;   // photons per workgroup of the sort
      int tx = 1, ty = 8;   // the squarest of the four tilings (ties: more cuts in y, whose rows are further apart in memory)
      for (int cx = 2; cx <= 8; cx *= 2)
        if ((double)h->nx / cx + (double)h->ny / (8 / cx) < (double)h->nx / tx + (double)h->ny / ty) { tx = cx; ty = 8 / cx; }
      hipLaunchKernelGGL(slab_count_kernel, dim3(kSlabSortBlocks), dim3(256), 0, h->stream, A.seed0, A.seed1, A.firstPhoton, A.nPhotons, span,
                         tx, ty, (unsigned *)sb.blockCounts.p);
      hipLaunchKernelGGL(slab_scan_kernel, dim3(1), dim3(8), 0, h->stream, (int)kSlabSortBlocks, (const unsigned *)sb.blockCounts.p,
                         (SlabMeta *)sb.meta.p, (unsigned *)sb.blockBase.p);
      hipLaunchKernelGGL(slab_fill_kernel, dim3(kSlabSortBlocks), dim3(256), 0, h->stream, A.seed0, A.seed1, A.firstPhoton, A.nPhotons, span,
                         tx, ty, (const unsigned *)sb.blockBase.p, (uint32_t *)sb.ids.p);
      HIPCHK(h, hipGetLastError());
      B.slabIds = (const uint32_t *)sb.ids.p; B.slabMeta = (SlabMeta *)sb.meta.p;
    }
  }
  // at most 256 photons per visit of the work counter: four photon generations of a wave.  Measured (I3RC_CHUNK_MAX, a
  // tuning knob): 128 loses a third (a returning atomic every other generation), 256...448 are equal, 1024 loses
  // 0.5 % on the step cloud and 2-5 % on the radar / Landsat cases to the imbalance at the end of a launch.
  static const long long chunkMax = std::getenv("I3RC_CHUNK_MAX") ? std::max(64ll, std::atoll(std::getenv("I3RC_CHUNK_MAX"))) : 256;
  B.chunk = (int)std::min<long long>(chunkMax, std::max<long long>(64, A.nPhotons / (blocks * (threads / 64) * 8)));
  HIPCHK(h, hipMemsetAsync(A.workCounter, 0, sizeof(unsigned long long), h->stream));
  const int slot = (int)(h->timedLaunches % i3rc_hip_integrator::kEventRing);
  if (timeIt) HIPCHK(h, hipEventRecord(h->evStart[slot], h->stream));
  {
    // thresholds the caller did not fix are adapted per wave (photon_kernel); negative = adaptive, starting value
    const int evThreshold = h->evThreshold > 0 ? h->evThreshold : -40;
    const int lightThreshold = h->lightThreshold > 0 ? h->lightThreshold : -24;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(threads), ldsBytes, h->stream, plan.P, B, evThreshold, lightThreshold);
  }
  HIPCHK(h, hipGetLastError());
  if (timeIt) { HIPCHK(h, hipEventRecord(h->evStop[slot], h->stream)); h->timedLaunches++; }
  return absorbed_columns(h, h->stream, plan.P.tally, 1, 0);
}


// ---- fused multi-batch launches ---------------------------------------------------------------------------------------
// out[b][e] = sum over the replicas r of blocks[b * R + r][e]
__global__ void __launch_bounds__(256) reduce_replicas_kernel(const double *blocks, double *out, long long nBlocksOut, int R, long long stride) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nBlocksOut * stride) return;
  const long long b = i / stride, e = i - b * stride;
  const double *src = blocks + (size_t)b * R * stride + e;
  double v = 0.0;
  for (int r = 0; r < R; ++r) v += src[(size_t)r * stride];
  out[i] = v;
}

// counters of batch b of the group = sum over the copies of its counter block (RunArgs::counterBlocks), into the batch's tally block
__global__ void __launch_bounds__(256) reduce_counters_kernel(const double *counterBlocks, double *out, long long nBlocksOut, long long stride, int oCnt) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nBlocksOut * I3RC_NUM_COUNTERS) return;
  const long long b = i / I3RC_NUM_COUNTERS;
  const int k = (int)(i - b * I3RC_NUM_COUNTERS);
  const double *src = counterBlocks + (size_t)b * kCounterReplicas * I3RC_NUM_COUNTERS + k;
  double v = 0.0;
  for (int r = 0; r < kCounterReplicas; ++r) v += src[(size_t)r * I3RC_NUM_COUNTERS];
  out[(size_t)b * stride + oCnt + k] = v;
}

// ---- batch moments on the device (i3rc_hip_run_batches_moments) ---------------------------------------------------------
// A driver only ever wants the first two moments over its batches of what reportResults hands out (monteCarloDriver.f95:
// 300-321, reduced at :333-352): per batch the raw block is normalised exactly as i3rc_hip_normalise does it (:353-395, float64,
// rounded to the reference's real(4)) and x, x^2 are added up per element -- the per-batch blocks never leave the device.
struct MomentsDev {
  // the raw tally block (offsets as in DevProblem) ...
  int oUp, oDown, oAbs, oVol, oInt, oExc, oCnt;
  int nx, ny, nz, ncomp, nDir, xyRegular, limitContrib;
  const double *areaFrac;   // [ncol] irregular grids: column area / domain area (:358-366)
  const double *dz;         // [nz] layer depths (:378-381)
  // ... and the moments block (i3rc_moments_layout)
  long long mUp, mDown, mAbs, mVol, mInt, mProfile, mMeanUp, mMeanDown, mMeanAbs, mMeanInt;
};
__device__ __forceinline__ double photons_per_column(const MomentsDev &M, const double *raw, int col) {
  const double nPhot = raw[M.oCnt + I3RC_CNT_PHOTONS];
  return M.xyRegular ? nPhot / (double)(M.nx * M.ny) : M.areaFrac[col] * nPhot;
}
// the radiance of column `col` in direction d as reportResults gives it: components summed, the excess of limited contributions
// redistributed in proportion (:327-347; excessSums[(j * nDir + d)] = the sum over the columns of component j's field)
__device__ __forceinline__ float normalised_intensity(const MomentsDev &M, const double *raw, const double *excessSums, int d, int col) {
  const size_t ncol = (size_t)M.nx * M.ny;
  double tot = 0.0;
  for (int j = 0; j <= M.ncomp; ++j) tot += raw[M.oInt + ((size_t)j * M.nDir + d) * ncol + col];
  if (M.limitContrib)
    for (int j = 0; j <= M.ncomp; ++j) {
      const double ex = raw[M.oExc + (size_t)j * M.nDir + d];
      if (ex > 0.0) tot += (raw[M.oInt + ((size_t)j * M.nDir + d) * ncol + col] / excessSums[j * M.nDir + d]) * ex;
    }
  return (float)(tot / photons_per_column(M, raw, col));
}
// one normalised field value of a batch; e counts through fluxUp | fluxDown | fluxAbsorbed | volumeAbsorption | intensity
__device__ __forceinline__ float normalised_value(const MomentsDev &M, const double *raw, const double *excessSums, long long e) {
  const long long ncol = (long long)M.nx * M.ny, ncell = ncol * M.nz;
  if (e < 3 * ncol) {
    const int which = (int)(e / ncol), col = (int)(e - which * ncol);
    const int o = which == 0 ? M.oUp : (which == 1 ? M.oDown : M.oAbs);
    return (float)(raw[o + col] / photons_per_column(M, raw, col));
  }
  e -= 3 * ncol;
  if (e < ncell) {
    const int kz = (int)(e / ncol), col = (int)(e - kz * ncol);
    return (float)(raw[M.oVol + e] / (photons_per_column(M, raw, col) * M.dz[kz]));
  }
  e -= ncell;
  const int d = (int)(e / ncol);
  return normalised_intensity(M, raw, excessSums, d, (int)(e - d * ncol));
}
// excessSums[b][(j * nDir + d)] = sum over the columns of intensityByComponent(:, :, d, j) of batch b (one workgroup each)
__global__ void __launch_bounds__(256) moments_excess_kernel(MomentsDev M, const double *blocks, long long stride, double *excessSums) {
  const int per = (M.ncomp + 1) * M.nDir, b = blockIdx.x / per, jd = blockIdx.x - b * per;
  const size_t ncol = (size_t)M.nx * M.ny;
  const double *f = blocks + (size_t)b * stride + M.oInt + (size_t)jd * ncol;
  double v = 0.0;
  for (size_t k = threadIdx.x; k < ncol; k += 256) v += f[k];
  __shared__ double part[4];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) excessSums[(size_t)b * per + jd] = part[0] + part[1] + part[2] + part[3];
}
// fields: one thread per element and SLICE of the group's batches (blockIdx.y; a slice's batches one after the other).  A small domain
// with many batches -- 700 elements x 1e5 batches of a thousand photons -- would otherwise be a few hundred threads walking 1e5 batches
// each, longer than the trace itself (round-4 advisor): the launch cuts the batches into as many slices as fill the chip.
__global__ void __launch_bounds__(256) moments_fields_kernel(MomentsDev M, const double *blocks, int count, long long stride, const double *excessSums,
                                                             long long nFields, double *sum, double *sumSq) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= nFields) return;
  const int per = (M.ncomp + 1) * M.nDir;
  const int perSlice = (count + (int)gridDim.y - 1) / (int)gridDim.y;
  const int b0 = (int)blockIdx.y * perSlice, b1 = min(count, b0 + perSlice);
  double s1 = 0.0, s2 = 0.0;
  for (int b = b0; b < b1; ++b) {
    const double x = (double)normalised_value(M, blocks + (size_t)b * stride, excessSums + (size_t)b * per, e);
    s1 += x; s2 += x * x;
  }
  // (fields lie in the moments block in the order e counts them: fluxUp | fluxDown | fluxAbsorbed | volumeAbsorption | intensity)
  unsafeAtomicAdd(sum + M.mUp + e, s1);
  unsafeAtomicAdd(sumSq + M.mUp + e, s2);
}
// domain means (:739-742), the absorption profile (:780) and the mean radiances: one workgroup per (batch, quantity)
__global__ void __launch_bounds__(256) moments_means_kernel(MomentsDev M, const double *blocks, long long stride, const double *excessSums,
                                                            double *sum, double *sumSq, double *counterTotals) {
  const int nq = 3 + M.nz + M.nDir, b = blockIdx.x / nq, q = blockIdx.x - b * nq;
  const double *raw = blocks + (size_t)b * stride;
  const int per = (M.ncomp + 1) * M.nDir;
  const long long ncol = (long long)M.nx * M.ny;
  long long first, at;   // first field element of the quantity (normalised_value's numbering) and its place in the moments block
  if (q < 3) { first = q * ncol; at = q == 0 ? M.mMeanUp : (q == 1 ? M.mMeanDown : M.mMeanAbs); }
  else if (q < 3 + M.nz) { first = 3 * ncol + (long long)(q - 3) * ncol; at = M.mProfile + (q - 3); }
  else { first = 3 * ncol + ncol * M.nz + (long long)(q - 3 - M.nz) * ncol; at = M.mMeanInt + (q - 3 - M.nz); }
  double v = 0.0;
  for (long long k = threadIdx.x; k < ncol; k += 256) v += (double)normalised_value(M, raw, excessSums + (size_t)b * per, first + k);
  __shared__ double part[4];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double x = (double)(float)((part[0] + part[1] + part[2] + part[3]) / (double)ncol);
    unsafeAtomicAdd(sum + at, x);
    unsafeAtomicAdd(sumSq + at, x * x);
  }
  if (q == 0 && threadIdx.x < I3RC_NUM_COUNTERS) unsafeAtomicAdd(counterTotals + threadIdx.x, raw[M.oCnt + threadIdx.x]);
}

// Can the batches of a driver's loop share one grid?  The specialised kernels -- the common problem class, production streams --,
// with or (round 4) without radiance directions: a local-estimate ray carries its batch in its info word (photon_kernel).
bool fusable(const i3rc_hip_integrator *h, int64_t nPhotons) {
  static const bool envOff = std::getenv("I3RC_FUSED") && std::atoi(std::getenv("I3RC_FUSED")) == 0;
  if (envOff || h->fusion == 0) return false;
  // (a batch's tally block beyond 256 MiB -- 3e7 cells -- would make a slot's pinned copy and its blocks unreasonably large: such
  // domains keep one launch per batch, whose tail is a small part of a launch that long anyway)
  static const bool radianceOff = std::getenv("I3RC_FUSED_RADIANCE") && std::atoi(std::getenv("I3RC_FUSED_RADIANCE")) == 0;
  if (h->nDir > 0 && (radianceOff || kNestedBuild)) return false;
  // (round 5: the widened class too -- several components, an irregular x / y grid, a gridded surface: photon_kernel<PhiloxBatchStream, ..., MULTI>)
  return (common_class(h, 0) || multi_class(h, 0)) && h->kernelVariant != I3RC_KERNEL_GENERAL && nPhotons < ((int64_t)1 << 31) &&
         h->layout.total * (int64_t)sizeof(double) <= ((int64_t)256 << 20);
}

// Replicas of a batch's tally block: enough that the hot words of a small domain (fluxUp / fluxDown of nx * ny columns) are
// spread over some 256 cache lines -- measured (tools/microbench/atomic_rate.hip): 64 hot float64 in one block take 4.1e8
// atomics/s from the whole chip, in 8 blocks 1.7e9, in 64 blocks 1.3e10; the step cloud needs 3.4e9.
int fused_replicas(const i3rc_hip_integrator *h) {
  const int64_t ncol = (int64_t)h->nx * h->ny, hotLines = 2 * ((ncol + 15) / 16);
  return (int)std::max<int64_t>(1, std::min<int64_t>(64, 256 / hotLines));
}

// Batches per group: device memory for the blocks (<= 1 GiB) and the pinned copy (<= 256 MiB) bound it; beyond that a
// group wants some 2.5e8 photons: a group boundary costs about 2.4 ms even with the next group queued behind it (step
// cloud, 300 batches of 1e6 photons: 15 groups 0.38, 6 groups 0.34, 3 groups 0.33, 1 group 0.31 ms per batch).
int fused_group_size(const i3rc_hip_integrator *h, int nBatches, int64_t nPhotons, bool moments = false) {
  const int64_t blockBytes = h->layout.total * 8, R = fused_replicas(h);
  int64_t g = ((int64_t)1 << 30) / (blockBytes * R);
  if (!moments) g = std::min<int64_t>(g, ((int64_t)256 << 20) / blockBytes);   // (moments mode: no pinned copy of the blocks)
  static const int64_t target = std::getenv("I3RC_FUSED_GROUP_PHOTONS") ? std::atoll(std::getenv("I3RC_FUSED_GROUP_PHOTONS")) : 250000000ll;
  g = std::min<int64_t>(g, (target + nPhotons - 1) / nPhotons);
  if (h->nDir > 0) g = std::min<int64_t>(g, 8192);   // (a local-estimate ray carries its batch in 13 bits of its info word: photon_kernel, make_ray)
  return (int)std::max<int64_t>(1, std::min<int64_t>(g, nBatches));
}

// (`reserve`: batches the slot's buffers are made for at least -- a slot that had to grow later would hipFree / hipMalloc while
// other groups are under way, and both wait for the device: a driver's look-ahead, whose groups grow 8, 16 ... 256, stood still
// for a whole group each time)
int ready_fused_slot(i3rc_hip_integrator *h, i3rc_hip_integrator::FusedSlot &g, int count, int R, int reserve, bool toHost) {
  count = std::max(count, reserve);
  const size_t outBytes = (size_t)count * h->layout.total * sizeof(double);
  // All groups share ONE stream (the first slot's): a kernel trace of groups on streams of their own showed consecutive
  // launches overlapping by 2 ms only -- the next kernel gets its first workgroup slots when the one before drains -- while
  // every launch took 5 ms longer than it does alone: with work pending in a second hardware queue the persistent kernel's
  // waves are time-sliced against it.  One queue exposes a launch's 1.4 ms tail per group and nothing else.
  static const bool ownStreams = std::getenv("I3RC_FUSED_STREAMS") && std::atoi(std::getenv("I3RC_FUSED_STREAMS")) != 0;
  if (!g.stream) {
    if (ownStreams || &g == &h->fused[0]) HIPCHK(h, hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    else {
      if (!h->fused[0].stream) HIPCHK(h, hipStreamCreateWithFlags(&h->fused[0].stream, hipStreamNonBlocking));
      g.stream = h->fused[0].stream;
    }
  }
  if (!g.done) HIPCHK(h, hipEventCreateWithFlags(&g.done, hipEventDisableTiming));
  if (!g.traced) HIPCHK(h, hipEventCreateWithFlags(&g.traced, hipEventDisableTiming));
  if (!h->fusedCopyStream) HIPCHK(h, hipStreamCreateWithFlags(&h->fusedCopyStream, hipStreamNonBlocking));
  if (!g.counter.p) HIPCHK(h, g.counter.alloc(sizeof(unsigned long long)));
  if (!g.abortFlag) HIPCHK(h, hipHostMalloc((void **)&g.abortFlag, sizeof(int), hipHostMallocCoherent | hipHostMallocMapped));
  if (toHost && g.pinnedBytes < outBytes) {
    if (g.pinned) { HIPCHK(h, hipHostFree(g.pinned)); g.pinned = nullptr; g.pinnedBytes = 0; }
    HIPCHK(h, hipHostMalloc((void **)&g.pinned, outBytes, hipHostMallocDefault));
    g.pinnedBytes = outBytes;
  }
  if (g.blocks.bytes < outBytes * R) HIPCHK(h, g.blocks.alloc(outBytes * R));
  if (R > 1 && g.compact.bytes < outBytes) HIPCHK(h, g.compact.alloc(outBytes));
  const size_t cntBytes = (size_t)count * kCounterReplicas * I3RC_NUM_COUNTERS * sizeof(double);
  if (g.counterBlocks.bytes < cntBytes) HIPCHK(h, g.counterBlocks.alloc(cntBytes));
  return 0;
}

// One group: `count` batches with the keys (seed0, seed1 .. seed1 + count - 1), nPhotons photons each, traced by ONE grid;
// zero, trace, sum the replicas, copy to the slot's pinned buffer -- all asynchronous on the slot's stream.
int accumulate_moments(i3rc_hip_integrator *h, hipStream_t stream, const double *blocks, int count, DevBuf &excess);

// (moments: the group's blocks stay on the device -- normalised and added to the handle's moment sums there, see accumulate_moments)
int launch_fused_group(i3rc_hip_integrator *h, i3rc_hip_integrator::FusedSlot &g, uint32_t seed0, uint32_t seed1, int count,
                       int64_t nPhotons, const i3rc_source *src, bool timeIt, int reserve = 0, bool moments = false) {
  const int R = fused_replicas(h);
  if (ready_fused_slot(h, g, count, R, reserve, !moments)) return 1;
  hipStream_t const callerStream = h->stream;
  double *const callerTally = h->tally;
  h->stream = g.stream; h->tally = (double *)g.blocks.p;   // (make_problem reads these two)
  LaunchPlan plan;
  RunArgs A;
  std::memset(&A, 0, sizeof(A));
  A.seed0 = seed0; A.seed1 = seed1; A.firstPhoton = 0; A.nPhotons = nPhotons;
  A.workCounter = (unsigned long long *)g.counter.p;
  int rc = make_problem(h, plan, true) || upload_source(h, src, nPhotons, A);
  h->stream = callerStream; h->tally = callerTally;
  if (rc) return 1;
  // (chunks: a wave takes this many photons of ONE batch per visit of the work counter; a lane hands its counts over when
  // its batch changes, so longer chunks mean fewer atomics, shorter ones a shorter end of the launch)
  static const int chunkEnv = std::getenv("I3RC_FUSED_CHUNK") ? std::max(64, std::atoi(std::getenv("I3RC_FUSED_CHUNK"))) : 0;
  A.chunk = chunkEnv > 0 ? chunkEnv : 512;
  if ((int64_t)A.chunk > nPhotons) A.chunk = (int)std::max<int64_t>(64, nPhotons);
  A.nBatches = (unsigned)count;
  A.chunksPerBatch = (unsigned)((nPhotons + A.chunk - 1) / A.chunk);
  A.replicas = R;
  A.blockStride = h->layout.total;
  A.abortFlag = g.abortFlag;
  A.counterBlocks = (double *)g.counterBlocks.p;
  if ((uint64_t)count * R >= ((uint64_t)1 << 31)) return h->fail("fused launch: too many tally blocks");
  using Kernel = void (*)(DevProblem, RunArgs, int, int);
  static const Kernel kernels[5] = {photon_kernel<PhiloxBatchStream, false, false, GRID_LDS>, photon_kernel<PhiloxBatchStream, false, false, GRID_GLOBAL>,
                                    photon_kernel<PhiloxBatchStream, false, false, GRID_BRICKS>, photon_kernel<PhiloxBatchStream, false, false, GRID_COLUMNS>, nullptr};
  const int place = plan.place;
  // the widened class (several components, irregular x / y, a gridded surface): its own fused kernels, flux ones too -- a driver's loop
  // of 1e6-photon batches on Landsat-36 + gas then costs 1.1 ms per batch instead of 2.6 (profiles/r05_fused_wide.txt)
  const bool wide = !common_class(h, 0);
  static const Kernel wideKernels[3][5] = {
      {photon_kernel<PhiloxBatchStream, false, false, GRID_LDS, false, false, true>, photon_kernel<PhiloxBatchStream, false, false, GRID_GLOBAL, false, false, true>, photon_kernel<PhiloxBatchStream, false, false, GRID_BRICKS, false, false, true>, photon_kernel<PhiloxBatchStream, false, false, GRID_COLUMNS, false, false, true>, photon_kernel<PhiloxBatchStream, false, false, GRID_COLBASE, false, false, true>},
      {photon_kernel<PhiloxBatchStream, true, false, GRID_LDS, false, false, true>, photon_kernel<PhiloxBatchStream, true, false, GRID_GLOBAL, false, false, true>, photon_kernel<PhiloxBatchStream, true, false, GRID_BRICKS, false, false, true>, photon_kernel<PhiloxBatchStream, true, false, GRID_COLUMNS, false, false, true>, photon_kernel<PhiloxBatchStream, true, false, GRID_COLBASE, false, false, true>},
      {photon_kernel<PhiloxBatchStream, true, false, GRID_LDS, false, true, true>, photon_kernel<PhiloxBatchStream, true, false, GRID_GLOBAL, false, true, true>, photon_kernel<PhiloxBatchStream, true, false, GRID_BRICKS, false, true, true>, photon_kernel<PhiloxBatchStream, true, false, GRID_COLUMNS, false, true, true>, photon_kernel<PhiloxBatchStream, true, false, GRID_COLBASE, false, true, true>}};
  // (the inverse table's cosines in LDS, workgroups of 1024 threads: as in launch(); these instantiations are planned for eight
  // waves per SIMD -- two workgroups per compute unit -- and pay for it with two vector registers in scratch)
  Kernel kern = kernels[place];
  int threads = 256;
  size_t ldsBytes = lds_bytes<PhiloxBatchStream>(h, plan, false);
  if (wide) kern = wideKernels[plan.intensity ? (direct_rays(h) ? 2 : 1) : 0][place];
  else
  if (plan.intensity) {   // radiance problems: through the event ring, or (one direction) without it -- as in launch()
    static const Kernel ring[4] = {photon_kernel<PhiloxBatchStream, true, false, GRID_LDS>, photon_kernel<PhiloxBatchStream, true, false, GRID_GLOBAL>,
                                   photon_kernel<PhiloxBatchStream, true, false, GRID_BRICKS>, photon_kernel<PhiloxBatchStream, true, false, GRID_COLUMNS>};
    static const Kernel direct[4] = {photon_kernel<PhiloxBatchStream, true, false, GRID_LDS, false, true>, photon_kernel<PhiloxBatchStream, true, false, GRID_GLOBAL, false, true>,
                                     photon_kernel<PhiloxBatchStream, true, false, GRID_BRICKS, false, true>, photon_kernel<PhiloxBatchStream, true, false, GRID_COLUMNS, false, true>};
    kern = direct_rays(h) ? direct[place] : ring[place];
  } else {
    static const bool tblOn = !(std::getenv("I3RC_TABLE_LDS") && std::atoi(std::getenv("I3RC_TABLE_LDS")) == 0);
    static const int tblPlaces = std::getenv("I3RC_FUSED_TABLE_LDS_PLACES") ? std::atoi(std::getenv("I3RC_FUSED_TABLE_LDS_PLACES")) : 11;   // (measured: Landsat-36 +13 %, radar 640 +12 %, step cloud +1.5 ... 3 % in the kernels' own time; on column records +1 ... 2.5 %)
    if (!wide && tblOn && ((tblPlaces >> place) & 1) && place != GRID_BRICKS && (plan.P.uniformPf >= 1 || h->nInvEntries[0] == 1) &&
        plan.ldsBytes + sizeof(float) * (size_t)plan.P.comp0.nInv <= 79 * 1024) {
      static const Kernel tbl[4] = {photon_kernel<PhiloxBatchStream, false, false, GRID_LDS, true>, photon_kernel<PhiloxBatchStream, false, false, GRID_GLOBAL, true>,
                                    nullptr, photon_kernel<PhiloxBatchStream, false, false, GRID_COLUMNS, true>};
      kern = tbl[place];
      threads = 1024;
      ldsBytes = lds_bytes<PhiloxBatchStream>(h, plan, true);
    }
  }
  if (ldsBytes > 160 * 1024 - 256) return h->fail("the launch needs more LDS than a compute unit has");
  if (!kern) return h->fail("internal: no fused kernel for this problem at this place of the extinction field");
  const void *fn = (const void *)kern;
  {
    static const char *const placeName[5] = {"GRID_LDS", "GRID_GLOBAL", "GRID_BRICKS", "GRID_COLUMNS", "GRID_COLBASE"};
    static thread_local char name[112];
    std::snprintf(name, sizeof(name), "photon_kernel<PhiloxBatchStream, %s, false, %s%s%s>", plan.intensity ? "true" : "false", placeName[place],
                  threads == 1024 ? ", table in LDS" : (plan.intensity && direct_rays(h) ? ", one direction" : ""), wide ? ", wide" : "");
    h->lastKernelName = name;
  }
  int perCU = h->blocksPerCU;
  if (perCU <= 0) {
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, threads, ldsBytes) != hipSuccess || occ < 1) occ = 2;
    perCU = std::min(occ, 8);
    if (place == GRID_BRICKS) perCU = std::min(perCU, ncell_bytes(h) > ((size_t)16 << 20) ? 4 : 5);   // (as in launch())
  }
  if (ldsBytes > 48 * 1024)
    HIPCHK(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
  long long blocks = (long long)h->numCU * perCU;
  const long long need = ((long long)count * nPhotons + threads - 1) / threads;
  if (blocks > need) blocks = std::max(1ll, need);
  const size_t outBytes = (size_t)count * h->layout.total * sizeof(double);
  *g.abortFlag = 0;
  HIPCHK(h, hipMemsetAsync(g.blocks.p, 0, outBytes * R, g.stream));
  HIPCHK(h, hipMemsetAsync(g.counter.p, 0, sizeof(unsigned long long), g.stream));
  HIPCHK(h, hipMemsetAsync(g.counterBlocks.p, 0, (size_t)count * kCounterReplicas * I3RC_NUM_COUNTERS * sizeof(double), g.stream));
  const int slot = (int)(h->timedLaunches % i3rc_hip_integrator::kEventRing);
  if (timeIt) HIPCHK(h, hipEventRecord(h->evStart[slot], g.stream));
  {
    const int evThreshold = h->evThreshold > 0 ? h->evThreshold : -40;
    const int lightThreshold = h->lightThreshold > 0 ? h->lightThreshold : -24;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(threads), ldsBytes, g.stream, plan.P, A, evThreshold, lightThreshold);
  }
  HIPCHK(h, hipGetLastError());
  if (timeIt) { HIPCHK(h, hipEventRecord(h->evStop[slot], g.stream)); h->timedLaunches++; }
  const double *result = (const double *)g.blocks.p;
  if (R > 1) {
    const long long n = (long long)count * h->layout.total;
    hipLaunchKernelGGL(reduce_replicas_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g.stream, (const double *)g.blocks.p,
                       (double *)g.compact.p, (long long)count, R, (long long)h->layout.total);
    HIPCHK(h, hipGetLastError());
    result = (const double *)g.compact.p;
  }
  if (absorbed_columns(h, g.stream, const_cast<double *>(result), count, (long long)h->layout.total)) return 1;
  {
    const long long n = (long long)count * I3RC_NUM_COUNTERS;
    hipLaunchKernelGGL(reduce_counters_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g.stream, (const double *)g.counterBlocks.p,
                       const_cast<double *>(result), (long long)count, (long long)h->layout.total, (int)h->layout.counters);
    HIPCHK(h, hipGetLastError());
  }
  if (moments) {
    if (accumulate_moments(h, g.stream, result, count, g.excess)) return 1;
    HIPCHK(h, hipEventRecord(g.done, g.stream));
  } else {
    static const bool copyInLine = std::getenv("I3RC_FUSED_COPY_INLINE") && std::atoi(std::getenv("I3RC_FUSED_COPY_INLINE")) != 0;   // (measurement knob: the copy on the groups' own stream, as before round 4)
    hipStream_t const cs = copyInLine ? g.stream : h->fusedCopyStream;
    HIPCHK(h, hipEventRecord(g.traced, g.stream));
    if (!copyInLine) HIPCHK(h, hipStreamWaitEvent(cs, g.traced, 0));
    HIPCHK(h, hipMemcpyAsync(g.pinned, result, outBytes, hipMemcpyDeviceToHost, cs));
    HIPCHK(h, hipEventRecord(g.done, cs));
  }
  g.count = count; g.seed1 = seed1; g.next = 0;
  if (tracing()) std::fprintf(stderr, "[i3rc %9.3f ms] fused group launched: seed words %u .. %u (%d batches of %lld photons, %d replicas, chunk %d)\n", trace_ms(),
                              seed1, seed1 + (unsigned)count - 1u, count, (long long)nPhotons, R, A.chunk);
  return 0;
}

void moments_layout(const i3rc_hip_integrator *h, i3rc_moments_layout &L) {
  const int64_t ncol = (int64_t)h->nx * h->ny, ncell = ncol * h->nz;
  int64_t o = 0;
  L.fluxUp = o; o += ncol; L.fluxDown = o; o += ncol; L.fluxAbsorbed = o; o += ncol;
  L.volumeAbsorption = o; o += ncell;
  L.intensity = o; o += (int64_t)h->nDir * ncol;
  L.absorbedProfile = o; o += h->nz;
  L.meanFluxUp = o++; L.meanFluxDown = o++; L.meanFluxAbsorbed = o++;
  L.meanIntensity = o; o += h->nDir;
  L.total = o;
}

// makes the handle's moment sums ready (zeroed) for a loop of batches
int begin_moments(i3rc_hip_integrator *h) {
  i3rc_moments_layout L;
  moments_layout(h, L);
  const size_t bytes = (size_t)L.total * sizeof(double);
  if (h->momSum.bytes != bytes) { HIPCHK(h, h->momSum.alloc(bytes)); HIPCHK(h, h->momSq.alloc(bytes)); }
  if (!h->momCounters.p) HIPCHK(h, h->momCounters.alloc(I3RC_NUM_COUNTERS * sizeof(double)));
  if (!h->momDz.p) {
    std::vector<double> dz((size_t)h->nz), area((size_t)h->nx * h->ny);
    for (int k = 0; k < h->nz; ++k) dz[k] = (double)h->zE[k + 1] - h->zE[k];
    const double ax = (double)h->xE.back() - h->xE.front(), ay = (double)h->yE.back() - h->yE.front();
    for (int j = 0; j < h->ny; ++j)
      for (int i = 0; i < h->nx; ++i)
        area[(size_t)j * h->nx + i] = (((double)h->yE[j + 1] - h->yE[j]) * ((double)h->xE[i + 1] - h->xE[i])) / (ax * ay);   // (as i3rc_hip_normalise)
    HIPCHK(h, h->momDz.upload(dz.data(), dz.size() * sizeof(double)));
    HIPCHK(h, h->momArea.upload(area.data(), area.size() * sizeof(double)));
  }
  HIPCHK(h, hipMemset(h->momSum.p, 0, bytes));
  HIPCHK(h, hipMemset(h->momSq.p, 0, bytes));
  HIPCHK(h, hipMemset(h->momCounters.p, 0, I3RC_NUM_COUNTERS * sizeof(double)));
  return 0;
}

// `count` raw tally blocks on the device (the handle's layout, counters filled in) -> normalised, x and x^2 added to the
// handle's moment sums; asynchronous on `stream` (blocks from several streams may be added at the same time: float64 atomics)
int accumulate_moments(i3rc_hip_integrator *h, hipStream_t stream, const double *blocks, int count, DevBuf &excess) {
  i3rc_moments_layout L;
  moments_layout(h, L);
  MomentsDev M;
  M.oUp = (int)h->layout.fluxUp; M.oDown = (int)h->layout.fluxDown; M.oAbs = (int)h->layout.fluxAbsorbed;
  M.oVol = (int)h->layout.volumeAbsorption; M.oInt = (int)h->layout.intensityByComponent; M.oExc = (int)h->layout.intensityExcess;
  M.oCnt = (int)h->layout.counters;
  M.nx = h->nx; M.ny = h->ny; M.nz = h->nz; M.ncomp = h->ncomp; M.nDir = h->nDir; M.xyRegular = h->xyRegular;
  M.limitContrib = h->nDir > 0 && h->params.limitIntensityContributions;
  M.areaFrac = (const double *)h->momArea.p; M.dz = (const double *)h->momDz.p;
  M.mUp = L.fluxUp; M.mDown = L.fluxDown; M.mAbs = L.fluxAbsorbed; M.mVol = L.volumeAbsorption; M.mInt = L.intensity;
  M.mProfile = L.absorbedProfile; M.mMeanUp = L.meanFluxUp; M.mMeanDown = L.meanFluxDown; M.mMeanAbs = L.meanFluxAbsorbed; M.mMeanInt = L.meanIntensity;
  const long long stride = h->layout.total;
  const int per = (h->ncomp + 1) * h->nDir;
  if (M.limitContrib) {
    const size_t need = (size_t)count * per * sizeof(double);
    if (excess.bytes < need) HIPCHK(h, excess.alloc(need));
    hipLaunchKernelGGL(moments_excess_kernel, dim3((unsigned)(count * per)), dim3(256), 0, stream, M, blocks, stride, (double *)excess.p);
  }
  const long long nFields = L.absorbedProfile;   // (fluxUp ... intensity: everything in front of the profile)
  // (slices of the batch range: enough threads for the chip -- 256 CUs x 8 workgroups -- however few the elements are, at least 8 batches a slice)
  const long long fieldBlocks = (nFields + 255) / 256;
  const int slices = (int)std::max<long long>(1, std::min<long long>({(long long)(count + 7) / 8, (2048 + fieldBlocks - 1) / fieldBlocks, 65535}));
  hipLaunchKernelGGL(moments_fields_kernel, dim3((unsigned)fieldBlocks, (unsigned)slices), dim3(256), 0, stream, M, blocks, count, stride,
                     (const double *)excess.p, nFields, (double *)h->momSum.p, (double *)h->momSq.p);
  hipLaunchKernelGGL(moments_means_kernel, dim3((unsigned)(count * (3 + h->nz + h->nDir))), dim3(256), 0, stream, M, blocks, stride,
                     (const double *)excess.p, (double *)h->momSum.p, (double *)h->momSq.p, (double *)h->momCounters.p);
  HIPCHK(h, hipGetLastError());
  return 0;
}

// (hostTallies == nullptr: moments mode -- nothing comes back per batch, see i3rc_hip_run_batches_moments)
int run_batches_fused(i3rc_hip_integrator *h, uint32_t seed0, uint32_t seed1, int nBatches, int64_t nPhotons, const i3rc_source *src,
                      double *hostTallies) {
  const bool moments = hostTallies == nullptr;
  const int G = fused_group_size(h, nBatches, nPhotons, moments);
  const size_t blockBytes = (size_t)h->layout.total * sizeof(double);
  drop_lookahead(h);
  HIPCHK(h, hipStreamSynchronize(h->stream));   // whatever the caller had in flight on the handle's stream comes first
  for (auto &g : h->fused) g.count = 0;
  int rc = 0;
  auto collect = [&](i3rc_hip_integrator::FusedSlot &g) -> int {
    if (g.count == 0) return 0;
    HIPCHK(h, hipEventSynchronize(g.done));
    if (!moments) std::memcpy(hostTallies + (size_t)g.first * (size_t)h->layout.total, g.pinned, (size_t)g.count * blockBytes);
    g.count = 0;
    return 0;
  };
  int k = 0;
  for (int first = 0; first < nBatches && !rc; first += G, ++k) {
    auto &g = h->fused[k % i3rc_hip_integrator::kFusedSlots];
    if ((rc = collect(g))) break;
    const int count = std::min(G, nBatches - first);
    rc = launch_fused_group(h, g, seed0, seed1 + (uint32_t)first, count, nPhotons, src, true, G, moments);
    if (!rc) g.first = first;
  }
  for (int j = 0; j < i3rc_hip_integrator::kFusedSlots; ++j) {   // drain in launch order (also after a failure: nothing stays in flight)
    auto &g = h->fused[(k + j) % i3rc_hip_integrator::kFusedSlots];
    if (rc) { if (g.stream) (void)hipStreamSynchronize(g.stream); g.count = 0; }
    else rc = collect(g);
  }
  return rc;
}


// Launches fused groups ahead of the caller until three are under way: the batches after the last group launched (or from
// `nextIfNone`), 8, 16, 32 ... 256 batches a group -- small groups first, so that the caller's first batches are there soon.
// A loop the caller has announced (i3rc_hip_expect_batches) is never overshot.
void top_up_groups(i3rc_hip_integrator *h, uint32_t seed0, uint32_t nextIfNone, int64_t nPhotons, const i3rc_source *src) {
  while ((int)h->aheadGroups.size() < i3rc_hip_integrator::kFusedSlots) {
    int k = -1;
    for (int j = 0; j < i3rc_hip_integrator::kFusedSlots; ++j) if (h->fused[j].count == 0) { k = j; break; }
    if (k < 0) break;
    uint32_t next = nextIfNone;
    if (!h->aheadGroups.empty()) { const auto &last = h->fused[h->aheadGroups.back()]; next = last.seed1 + (uint32_t)last.count; }
    int want = h->aheadGroupSize <= 0 ? 8 : std::min(256, 2 * h->aheadGroupSize);
    if (h->aheadBounded) {
      const int64_t left = (int64_t)h->aheadEnd - (int64_t)next;
      if (left <= 0) break;
      want = (int)std::min<int64_t>(want, left);
    }
    int size = fused_group_size(h, want, nPhotons);
    // What a slot is made for.  An ANNOUNCED loop: its largest group at once (a slot that grows later waits for the device twice).
    // A loop that is only guessed at (an unchanged driver's calls): at most 96 MiB of pinned memory per slot -- a Landsat-sized
    // field would otherwise hold 3 x (256 MiB pinned + 256 MiB of device memory) per handle for a loop of any length --, and its
    // groups stay within that, so that such a slot never grows either.
    int reserve;
    if (h->aheadBounded) reserve = (int)std::min<int64_t>(fused_group_size(h, 256, nPhotons), (int64_t)h->aheadEnd - (int64_t)next);
    else {
      const int64_t blockBytes = h->layout.total * (int64_t)sizeof(double);
      reserve = (int)std::max<int64_t>(1, std::min<int64_t>(fused_group_size(h, 256, nPhotons), ((int64_t)96 << 20) / blockBytes));
      size = std::min(size, reserve);
    }
    if (launch_fused_group(h, h->fused[k], seed0, next, size, nPhotons, src, false, std::max(reserve, size))) {   // (not the caller's failure)
      h->fused[k].count = 0;
      h->fusedAheadFailed = true;   // (remembered: the caller's loop goes on with single batches launched ahead, see i3rc_hip_compute_batch)
      break;
    }
    h->aheadGroupSize = std::max(h->aheadGroupSize, want);
    h->aheadGroups.push_back(k);
  }
}

}  // namespace

extern "C" {

int i3rc_hip_launch_batch(i3rc_hip_integrator *h, uint32_t seed0, uint32_t seed1, int64_t firstPhoton, int64_t nPhotons,
                          const i3rc_source *src) {
  if (!h) return 1;
  if (!src) return h->fail("i3rc_hip_launch_batch: null source");
  if (nPhotons <= 0) return h->fail("setIllumination: must ask for non-negative number of photons.");  // illumination :78-79
  HIPCHK(h, hipSetDevice(h->device));
  LaunchPlan plan;
  if (make_problem(h, plan)) return 1;
  RunArgs A;
  std::memset(&A, 0, sizeof(A));
  A.seed0 = seed0; A.seed1 = seed1; A.firstPhoton = firstPhoton; A.nPhotons = nPhotons;
  A.workCounter = (unsigned long long *)h->workCounter.p;
  if (upload_source(h, src, nPhotons, A)) return 1;
  // Workgroups keep their partial flux / radiance sums in float32 (LDS): a workgroup must not see so many photons
  // that a column's sum could leave the range where float32 still counts (2^24).  Per-photon random streams make a
  // long batch the same as several launches over consecutive photon ranges, so very long Directional batches are cut
  // into launches of at most 2^22 photons per compute unit (about 1e9 photons on an MI355X).
  const int64_t perLaunch = h->launchLimit > 0 ? h->launchLimit : (int64_t)h->numCU << 22;
  if (src->kind != 0 || nPhotons <= perLaunch) return launch<PhiloxStream>(h, plan, A, true);
  for (int64_t done = 0; done < nPhotons; done += perLaunch) {
    RunArgs part = A;
    part.firstPhoton = firstPhoton + done;
    part.nPhotons = std::min(perLaunch, nPhotons - done);
    if (launch<PhiloxStream>(h, plan, part, true)) return 1;
  }
  return 0;
}

// The batch loop of a driver (Example-Drivers/monteCarloDriver.f95:283-326) as one call: see include/i3rc_hip.h
int i3rc_hip_run_batches(i3rc_hip_integrator *h, uint32_t seed0, uint32_t seed1, int nBatches, int64_t nPhotons,
                         const i3rc_source *src, int inFlight, double *hostTallies) {
  if (!h) return 1;
  if (!src || !hostTallies || nBatches < 1) return h->fail("i3rc_hip_run_batches: bad arguments");
  if (src->kind != 0) return h->fail("i3rc_hip_run_batches: Directional photon streams only (explicit streams differ from batch to batch)");
  if (nPhotons <= 0) return h->fail("setIllumination: must ask for non-negative number of photons.");
  HIPCHK(h, hipSetDevice(h->device));
  // batches that the specialised flux kernels run share one grid, group by group (see FusedSlot); one long batch alone, or
  // batches of a size at which a launch's tail no longer matters, go one launch each as before
  if (fusable(h, nPhotons) && (h->fusion == 1 || (nBatches >= 2 && nPhotons <= 20000000)))
    return run_batches_fused(h, seed0, seed1, nBatches, nPhotons, src, hostTallies);
  const int K = std::min(nBatches, inFlight <= 0 ? 6 : std::min(inFlight, (int)i3rc_hip_integrator::kMaxInFlight));
  const size_t bytes = (size_t)h->layout.total * sizeof(double);
  drop_lookahead(h);               // (i3rc_hip_compute_batch shares the slots)
  if (reset_slots_if_layout_changed(h)) return 1;
  for (int k = 0; k < K; ++k) {
    if (ready_slot(h, k)) return 1;
    h->pipe[k].batch = -1;
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));   // whatever the caller had in flight on the handle's stream comes first
  hipStream_t const callerStream = h->stream;
  double *const callerTally = h->tally;
  int rc = 0;
  auto collect = [&](i3rc_hip_integrator::PipeSlot &sl) -> int {
    if (sl.batch < 0) return 0;
    HIPCHK(h, hipEventSynchronize(sl.done));
    std::memcpy(hostTallies + (size_t)sl.batch * (size_t)h->layout.total, sl.pinned, bytes);
    sl.batch = -1;
    return 0;
  };
  const int64_t perLaunch = h->launchLimit > 0 ? h->launchLimit : (int64_t)h->numCU << 22;
  for (int b = 0; b < nBatches && !rc; ++b) {
    auto &sl = h->pipe[b % K];
    if ((rc = collect(sl))) break;
    h->stream = sl.stream; h->tally = (double *)sl.tally.p;   // (make_problem and launch read these two)
    LaunchPlan plan;
    if ((rc = make_problem(h, plan))) break;
    RunArgs A;
    std::memset(&A, 0, sizeof(A));
    A.seed0 = seed0; A.seed1 = seed1 + (uint32_t)b; A.firstPhoton = 0; A.nPhotons = nPhotons;
    A.workCounter = (unsigned long long *)sl.counter.p;
    if ((rc = upload_source(h, src, nPhotons, A))) break;
    if (hipMemsetAsync(sl.tally.p, 0, bytes, sl.stream) != hipSuccess) { rc = h->fail("i3rc_hip_run_batches: clearing a tally buffer failed"); break; }
    for (int64_t done = 0; done < nPhotons && !rc; done += perLaunch) {   // (very long batches: see i3rc_hip_launch_batch)
      RunArgs part = A;
      part.firstPhoton = done;
      part.nPhotons = std::min(perLaunch, nPhotons - done);
      rc = launch<PhiloxStream>(h, plan, part, true);
    }
    if (rc) break;
    if (hipMemcpyAsync(sl.pinned, sl.tally.p, bytes, hipMemcpyDeviceToHost, sl.stream) != hipSuccess ||
        hipEventRecord(sl.done, sl.stream) != hipSuccess) { rc = h->fail("i3rc_hip_run_batches: copying a batch's tallies back failed"); break; }
    sl.batch = b;
  }
  for (int k = 0; k < K; ++k) {   // drain (also after a failure: nothing of this call stays in flight)
    if (rc) { (void)hipStreamSynchronize(h->pipe[k].stream); h->pipe[k].batch = -1; }
    else rc = collect(h->pipe[k]);
  }
  h->stream = callerStream; h->tally = callerTally;
  return rc;
}

int i3rc_hip_get_moments_layout(const i3rc_hip_integrator *h, i3rc_moments_layout *layout) {
  if (!h || !layout) return 1;
  moments_layout(h, *layout);
  return 0;
}

// A driver's batch loop with its statistics gathered on the device: see include/i3rc_hip.h
int i3rc_hip_run_batches_moments(i3rc_hip_integrator *h, uint32_t seed0, uint32_t seed1, int nBatches, int64_t nPhotons,
                                 const i3rc_source *src, double *sum, double *sumSquares, double *counters) {
  if (!h) return 1;
  if (!src || !sum || !sumSquares || nBatches < 1) return h->fail("i3rc_hip_run_batches_moments: bad arguments");
  if (src->kind != 0) return h->fail("i3rc_hip_run_batches_moments: Directional photon streams only (explicit streams differ from batch to batch)");
  if (nPhotons <= 0) return h->fail("setIllumination: must ask for non-negative number of photons.");
  HIPCHK(h, hipSetDevice(h->device));
  drop_lookahead(h);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (begin_moments(h)) return 1;
  int rc = 0;
  if (fusable(h, nPhotons) && (h->fusion == 1 || (nBatches >= 2 && nPhotons <= 20000000)))
    rc = run_batches_fused(h, seed0, seed1, nBatches, nPhotons, src, nullptr);
  else {
    // other problems (the general kernels; long batches): one launch per batch, several in flight as in i3rc_hip_run_batches, each
    // batch's block normalised and added up on its slot's stream behind its launch
    const int K = std::min(nBatches, 6);
    if (reset_slots_if_layout_changed(h)) return 1;
    for (int k = 0; k < K; ++k) { if (ready_slot(h, k)) return 1; h->pipe[k].batch = -1; }
    hipStream_t const callerStream = h->stream;
    double *const callerTally = h->tally;
    const size_t bytes = (size_t)h->layout.total * sizeof(double);
    const int64_t perLaunch = h->launchLimit > 0 ? h->launchLimit : (int64_t)h->numCU << 22;
    for (int b = 0; b < nBatches && !rc; ++b) {
      auto &sl = h->pipe[b % K];
      h->stream = sl.stream; h->tally = (double *)sl.tally.p;
      LaunchPlan plan;
      RunArgs A;
      std::memset(&A, 0, sizeof(A));
      A.seed0 = seed0; A.seed1 = seed1 + (uint32_t)b; A.firstPhoton = 0; A.nPhotons = nPhotons;
      A.workCounter = (unsigned long long *)sl.counter.p;
      rc = make_problem(h, plan) || upload_source(h, src, nPhotons, A);
      if (!rc && hipMemsetAsync(sl.tally.p, 0, bytes, sl.stream) != hipSuccess) rc = h->fail("i3rc_hip_run_batches_moments: clearing a tally buffer failed");
      for (int64_t done = 0; done < nPhotons && !rc; done += perLaunch) {
        RunArgs part = A;
        part.firstPhoton = done;
        part.nPhotons = std::min(perLaunch, nPhotons - done);
        rc = launch<PhiloxStream>(h, plan, part, true);
      }
      if (!rc) rc = accumulate_moments(h, sl.stream, (const double *)sl.tally.p, 1, sl.excess);
    }
    for (int k = 0; k < K; ++k) if (h->pipe[k].stream && hipStreamSynchronize(h->pipe[k].stream) != hipSuccess && !rc) rc = h->fail("i3rc_hip_run_batches_moments: waiting for the batches failed");
    h->stream = callerStream; h->tally = callerTally;
  }
  if (rc) return 1;
  i3rc_moments_layout L;
  moments_layout(h, L);
  HIPCHK(h, hipMemcpy(sum, h->momSum.p, (size_t)L.total * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(sumSquares, h->momSq.p, (size_t)L.total * sizeof(double), hipMemcpyDeviceToHost));
  if (counters) HIPCHK(h, hipMemcpy(counters, h->momCounters.p, I3RC_NUM_COUNTERS * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

// computeRadiativeTransfer for one batch of a driver's loop, looking ahead: see include/i3rc_hip.h
int i3rc_hip_compute_batch(i3rc_hip_integrator *h, uint32_t seed0, uint32_t seed1, int64_t nPhotons, const i3rc_source *src,
                           int lookAhead, double *hostTallies) {
  if (!h) return 1;
  if (!src || !hostTallies) return h->fail("i3rc_hip_compute_batch: bad arguments");
  if (src->kind != 0) return h->fail("i3rc_hip_compute_batch: Directional photon streams only");
  if (nPhotons <= 0) return h->fail("setIllumination: must ask for non-negative number of photons.");
  HIPCHK(h, hipSetDevice(h->device));
  const int depth = std::max(0, std::min(lookAhead, (int)i3rc_hip_integrator::kMaxInFlight - 1));
  const size_t bytes = (size_t)h->layout.total * sizeof(double);
  if (reset_slots_if_layout_changed(h)) return 1;   // (a change of the layout has dropped the queue already: set_directions)
  auto free_slot = [&]() -> int {
    for (int k = 0; k < i3rc_hip_integrator::kMaxInFlight; ++k) if (h->pipe[k].batch < 0) return k;
    return -1;
  };
  hipStream_t const callerStream = h->stream;
  double *const callerTally = h->tally;
  auto start = [&](int k, uint32_t s1, bool timed) -> int {   // zero, trace, copy back: asynchronous on the slot's stream (batches launched
                                                               // ahead of the caller are not recorded in the ring of timed launches)
    if (ready_slot(h, k)) return 1;
    auto &sl = h->pipe[k];
    h->stream = sl.stream; h->tally = (double *)sl.tally.p;
    int rc = 0;
    LaunchPlan plan;
    RunArgs A;
    std::memset(&A, 0, sizeof(A));
    A.seed0 = seed0; A.seed1 = s1; A.firstPhoton = 0; A.nPhotons = nPhotons;
    A.workCounter = (unsigned long long *)sl.counter.p;
    rc = make_problem(h, plan) || upload_source(h, src, nPhotons, A);
    if (!rc && hipMemsetAsync(sl.tally.p, 0, bytes, sl.stream) != hipSuccess) rc = h->fail("i3rc_hip_compute_batch: clearing a tally buffer failed");
    const int64_t perLaunch = h->launchLimit > 0 ? h->launchLimit : (int64_t)h->numCU << 22;
    for (int64_t done = 0; done < nPhotons && !rc; done += perLaunch) {
      RunArgs part = A;
      part.firstPhoton = done;
      part.nPhotons = std::min(perLaunch, nPhotons - done);
      rc = launch<PhiloxStream>(h, plan, part, timed);
    }
    if (!rc && (hipMemcpyAsync(sl.pinned, sl.tally.p, bytes, hipMemcpyDeviceToHost, sl.stream) != hipSuccess ||
                hipEventRecord(sl.done, sl.stream) != hipSuccess))
      rc = h->fail("i3rc_hip_compute_batch: copying a batch's tallies back failed");
    h->stream = callerStream; h->tally = callerTally;
    if (!rc) sl.batch = 0;   // in use (any value >= 0; i3rc_hip_run_batches keeps a batch number here)
    return rc;
  };
  i3rc_hip_integrator::BatchSignature sig;
  sig.seed0 = seed0; sig.n = nPhotons; sig.mu = src->solarMu; sig.az = src->solarAzimuth; sig.set = true;
  // Problems whose batches can share a grid (fusable) are looked ahead in GROUPS: 8, 16, 32 ... 256 batches per fused
  // launch, up to three groups under way (one after the other on the device), the caller served from the oldest.  Such launches are not timed (the ring of
  // i3rc_hip_kernel_ms_history holds launches the caller asked for), and a group that is not wanted after all is called
  // off through its abort word: its waves stop at their next visit of the work counter.
  const bool fuse = depth > 0 && fusable(h, nPhotons) && (h->fusion == 1 || nPhotons <= 20000000);
  // (a group that could not be launched -- memory -- is not tried again at every call: groups under way are still handed out)
  auto top_up = [&]() { if (!h->fusedAheadFailed) top_up_groups(h, seed0, seed1 + 1u, nPhotons, src); };
  if (!h->aheadGroups.empty()) {
    auto &g = h->fused[h->aheadGroups.front()];
    if (fuse && h->aheadSig == sig && g.seed1 + (uint32_t)g.next == seed1) {
      const double tw = tracing() ? trace_ms() : 0.0;
      if (hipEventSynchronize(g.done) != hipSuccess) { drop_lookahead(h); return h->fail("i3rc_hip_compute_batch: waiting for the batch failed"); }
      if (tracing() && trace_ms() - tw > 0.5) std::fprintf(stderr, "[i3rc %9.3f ms] batch %u waited %.3f ms for its group\n", trace_ms(), seed1, trace_ms() - tw);
      std::memcpy(hostTallies, g.pinned + (size_t)g.next * (size_t)h->layout.total, bytes);
      if (++g.next == g.count) { g.count = 0; h->aheadGroups.erase(h->aheadGroups.begin()); }
      h->lastSig = sig; h->lastSeed1 = seed1;
      top_up();
      if (h->aheadBounded && h->aheadGroups.empty()) h->aheadBounded = false;   // the announced loop is through: from here on the usual guessing
      return 0;
    }
    drop_lookahead(h);
  }
  // batches launched ahead serve this call only if the next of them is exactly this batch
  if (!h->aheadQueue.empty() && !(h->aheadSig == sig && h->aheadQueue.front().seed1 == seed1)) drop_lookahead(h);
  int mine;
  if (!h->aheadQueue.empty()) {
    mine = h->aheadQueue.front().slot;
    h->aheadQueue.erase(h->aheadQueue.begin());
  } else {
    mine = free_slot();
    if (mine < 0) return h->fail("i3rc_hip_compute_batch: no free slot");
    if (start(mine, seed1, true)) { drop_lookahead(h); return 1; }
  }
  h->pipe[mine].batch = 0;   // (in use until its tallies have been handed over)
  // look ahead once the caller's loop shows: the same batch as last time with the next seed word (monteCarloDriver.f95:277)
  const bool inLoop = h->lastSig == sig && h->lastSeed1 + 1u == seed1;
  h->lastSig = sig; h->lastSeed1 = seed1;
  if (fuse && inLoop) {
    h->aheadSig = sig;
    top_up();
  }
  if (!(fuse && inLoop && !h->aheadGroups.empty()) && depth > 0 && (inLoop || !h->aheadQueue.empty())) {   // (also when no fused group could be launched)
    h->aheadSig = sig;
    while ((int)h->aheadQueue.size() < depth) {
      const uint32_t next = (h->aheadQueue.empty() ? seed1 : h->aheadQueue.back().seed1) + 1u;
      const int k = free_slot();
      if (k < 0 || start(k, next, false)) break;   // (a failed look-ahead is not this batch's failure: the error text stays for the next call)
      h->aheadQueue.push_back({next, k});
    }
  }
  auto &sl = h->pipe[mine];
  if (hipEventSynchronize(sl.done) != hipSuccess) { drop_lookahead(h); sl.batch = -1; return h->fail("i3rc_hip_compute_batch: waiting for the batch failed"); }
  std::memcpy(hostTallies, sl.pinned, bytes);
  sl.batch = -1;
  return 0;
}

// Announces a driver's loop to i3rc_hip_compute_batch: see include/i3rc_hip.h
int i3rc_hip_expect_batches(i3rc_hip_integrator *h, uint32_t seed0, uint32_t seed1, int nBatches, int64_t nPhotons, const i3rc_source *src,
                            int *accepted) {
  if (!h) return 1;
  if (accepted) *accepted = 0;
  if (!src || nBatches < 1) return h->fail("i3rc_hip_expect_batches: bad arguments");
  if (src->kind != 0) return h->fail("i3rc_hip_expect_batches: Directional photon streams only");
  if (nPhotons <= 0) return h->fail("setIllumination: must ask for non-negative number of photons.");
  HIPCHK(h, hipSetDevice(h->device));
  drop_lookahead(h);
  if (!(fusable(h, nPhotons) && (h->fusion == 1 || (nBatches >= 2 && nPhotons <= 20000000)))) return 0;   // (the caller goes on as it would have)
  {   // the problem must be complete before anything is launched (as i3rc_hip_compute_batch would find out)
    LaunchPlan plan;
    if (make_problem(h, plan, true)) return 1;
  }
  i3rc_hip_integrator::BatchSignature sig;
  sig.seed0 = seed0; sig.n = nPhotons; sig.mu = src->solarMu; sig.az = src->solarAzimuth; sig.set = true;
  h->aheadSig = sig;
  h->lastSig = sig; h->lastSeed1 = seed1 - 1u;
  h->aheadBounded = true; h->aheadEnd = seed1 + (uint32_t)nBatches;
  h->aheadGroupSize = 16;   // (the first group: 32 batches)
  top_up_groups(h, seed0, seed1, nPhotons, src);
  // (the first group could not be launched -- the slot's pinned or device reserve, say: launch_fused_group has left the reason in
  // i3rc_hip_last_error --: not accepted, and the caller goes on with i3rc_hip_run_batches, as the header promises)
  if (h->aheadGroups.empty()) { drop_lookahead(h); return 0; }
  if (accepted) *accepted = 1;
  return 0;
}

int i3rc_hip_run_replay(i3rc_hip_integrator *h, int64_t nPhotons, const i3rc_source *src, const float *randoms,
                        int64_t nRandoms, const int64_t *drawStart, int32_t *fate, int32_t *fateColumn, float *fateWeight,
                        int32_t *fateOrder, int32_t *drawsUsed) {
  if (!h) return 1;
  if (!src || !randoms || !drawStart || nPhotons <= 0) return h->fail("i3rc_hip_run_replay: bad arguments");
  HIPCHK(h, hipSetDevice(h->device));
  LaunchPlan plan;
  if (make_problem(h, plan, false, true)) return 1;
  RunArgs A;
  std::memset(&A, 0, sizeof(A));
  A.nPhotons = nPhotons;
  A.workCounter = (unsigned long long *)h->workCounter.p;
  if (upload_source(h, src, nPhotons, A)) return 1;
  DevBuf dR, dS, dFate, dCol, dW, dOrd, dUsed;
  HIPCHK(h, dR.upload(randoms, sizeof(float) * (size_t)nRandoms));
  HIPCHK(h, dS.upload(drawStart, sizeof(int64_t) * (size_t)nPhotons));
  A.randoms = (const float *)dR.p; A.nRandoms = nRandoms; A.drawStart = (const long long *)dS.p;
  const bool rec = fate && fateColumn && fateWeight && fateOrder && drawsUsed;
  if (rec) {
    HIPCHK(h, dFate.alloc(sizeof(int32_t) * nPhotons)); HIPCHK(h, dCol.alloc(sizeof(int32_t) * nPhotons));
    HIPCHK(h, dW.alloc(sizeof(float) * nPhotons)); HIPCHK(h, dOrd.alloc(sizeof(int32_t) * nPhotons));
    HIPCHK(h, dUsed.alloc(sizeof(int32_t) * nPhotons));
    A.fate = (int32_t *)dFate.p; A.fateColumn = (int32_t *)dCol.p; A.fateWeight = (float *)dW.p;
    A.fateOrder = (int32_t *)dOrd.p; A.drawsUsed = (int32_t *)dUsed.p;
  }
  if (launch<ReplayStream>(h, plan, A, false)) return 1;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (rec) {
    HIPCHK(h, hipMemcpy(fate, dFate.p, sizeof(int32_t) * nPhotons, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(fateColumn, dCol.p, sizeof(int32_t) * nPhotons, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(fateWeight, dW.p, sizeof(float) * nPhotons, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(fateOrder, dOrd.p, sizeof(int32_t) * nPhotons, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(drawsUsed, dUsed.p, sizeof(int32_t) * nPhotons, hipMemcpyDeviceToHost));
  }
  return 0;
}

int i3rc_hip_trace_rays(i3rc_hip_integrator *h, int64_t n, const float *dir, float *pos, int32_t *idx, const float *target,
                        float *tau, int32_t *steps) {
  if (!h) return 1;
  if (n <= 0 || !dir || !pos || !idx || !target || !tau || !steps) return h->fail("i3rc_hip_trace_rays: bad arguments");
  for (int64_t i = 0; i < n; ++i) {  // shapes must match what the kernel indexes
    if (idx[3 * i] < 1 || idx[3 * i] > h->nx || idx[3 * i + 1] < 1 || idx[3 * i + 1] > h->ny || idx[3 * i + 2] < 1 ||
        idx[3 * i + 2] > h->nz)
      return h->fail("i3rc_hip_trace_rays: start cell outside the domain");
  }
  HIPCHK(h, hipSetDevice(h->device));
  LaunchPlan plan;
  // tables are not needed for bare tracing: build the problem without the completeness checks
  const std::vector<CompTables> saved = h->comp;
  static const float dummy = 0.f;
  for (int c = 0; c < h->ncomp; ++c) if (!h->comp[c].inv) h->comp[c].inv = &dummy;
  const int savedDir = h->nDir;
  h->nDir = 0;
  const int rc = make_problem(h, plan);
  h->nDir = savedDir;
  h->comp = saved;
  h->compDirty = true;
  if (rc) return 1;
  plan.P.ldsGrid = 0; plan.P.ldsTallies = 0;
  // the hook reads the bricked copy (its index is checked bit for bit) unless i3rc_hip_select_grid_place asked for the plain
  // field or the column records
  const int hookPlace = h->gridPlace == I3RC_GRID_LINEAR ? GRID_GLOBAL : (h->gridPlace == I3RC_GRID_COLUMNS ? (h->dColBase.p ? GRID_COLBASE : GRID_COLUMNS) : GRID_BRICKS);
  plan.P.extBrick = (const float *)h->dExtBrick.p;
  plan.P.colRec = (const uint2 *)h->dColRec.p;
  plan.P.colBase = (const float *)h->dColBase.p;
  DevBuf dDir, dPos, dIdx, dTar, dTau, dSteps;
  HIPCHK(h, dDir.upload(dir, sizeof(float) * 3 * n)); HIPCHK(h, dPos.upload(pos, sizeof(float) * 3 * n));
  HIPCHK(h, dIdx.upload(idx, sizeof(int32_t) * 3 * n)); HIPCHK(h, dTar.upload(target, sizeof(float) * n));
  HIPCHK(h, dTau.alloc(sizeof(float) * n)); HIPCHK(h, dSteps.alloc(sizeof(int32_t) * n));
  const size_t lds = sizeof(float) * ((h->nx + 1) + (h->ny + 1) + (h->nz + 1)) + sizeof(uint32_t) * std::max((size_t)h->clearWords, (size_t)h->nz);
  if (lds > 64 * 1024) return h->fail("i3rc_hip_trace_rays: domain edge vectors do not fit in LDS");
  // (more than 65534 layers: the clear-air map's 16-bit layer numbers do not reach the top -- the hook reads the bricks without it)
  auto *const hook = hookPlace == GRID_GLOBAL ? trace_rays_kernel<GRID_GLOBAL, false>
                   : (hookPlace == GRID_COLUMNS ? trace_rays_kernel<GRID_COLUMNS, false> : hookPlace == GRID_COLBASE ? trace_rays_kernel<GRID_COLBASE, false>
                                                : (h->nz <= 65534 ? trace_rays_kernel<GRID_BRICKS, true> : trace_rays_kernel<GRID_BRICKS, false>));
  hipLaunchKernelGGL(hook, dim3((unsigned)((n + 255) / 256)), dim3(256), lds, h->stream, plan.P, (long long)n,
                     (const float *)dDir.p, (float *)dPos.p, (int32_t *)dIdx.p, (const float *)dTar.p, (float *)dTau.p,
                     (int32_t *)dSteps.p);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(pos, dPos.p, sizeof(float) * 3 * n, hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(idx, dIdx.p, sizeof(int32_t) * 3 * n, hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(tau, dTau.p, sizeof(float) * n, hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(steps, dSteps.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
  return 0;
}

/* Test hook: raw Philox blocks and the float deviates derived from them, as photon streams see them. */
int i3rc_hip_philox_blocks(i3rc_hip_integrator *h, uint32_t seed0, uint32_t seed1, int64_t firstPhoton, int64_t n,
                           int blocksPerPhoton, uint32_t *out, float *outf) {
  if (!h) return 1;
  if (n <= 0 || blocksPerPhoton <= 0 || !out || !outf) return h->fail("i3rc_hip_philox_blocks: bad arguments");
  HIPCHK(h, hipSetDevice(h->device));
  DevBuf d, df;
  const size_t cnt = (size_t)n * blocksPerPhoton * 4;
  HIPCHK(h, d.alloc(cnt * 4)); HIPCHK(h, df.alloc(cnt * 4));
  hipLaunchKernelGGL(philox_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, seed0, seed1,
                     (long long)firstPhoton, (long long)n, blocksPerPhoton, (uint32_t *)d.p, (float *)df.p);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(out, d.p, cnt * 4, hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(outf, df.p, cnt * 4, hipMemcpyDeviceToHost));
  return 0;
}

/* Test hook: counts, over n pairs, where the kernel's exact_div / exact_sqrt differ from IEEE `/` and sqrtf. */
int i3rc_hip_arith_check(i3rc_hip_integrator *h, int64_t n, const float *num, const float *den, int64_t *divMismatch,
                         int64_t *sqrtMismatch) {
  if (!h) return 1;
  if (n <= 0 || !num || !den || !divMismatch || !sqrtMismatch) return h->fail("i3rc_hip_arith_check: bad arguments");
  HIPCHK(h, hipSetDevice(h->device));
  DevBuf dn, dd, dc;
  HIPCHK(h, dn.upload(num, sizeof(float) * n)); HIPCHK(h, dd.upload(den, sizeof(float) * n));
  HIPCHK(h, dc.alloc(2 * sizeof(unsigned long long)));
  HIPCHK(h, hipMemset(dc.p, 0, 2 * sizeof(unsigned long long)));
  hipLaunchKernelGGL(arith_check_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, (long long)n,
                     (const float *)dn.p, (const float *)dd.p, (unsigned long long *)dc.p);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  unsigned long long c[2];
  HIPCHK(h, hipMemcpy(c, dc.p, sizeof(c), hipMemcpyDeviceToHost));
  *divMismatch = (int64_t)c[0]; *sqrtMismatch = (int64_t)c[1];
  return 0;
}

/* Test hook: findIndex (Code/numericUtilities.f95:195-248) on the device, as the kernels evaluate it. */
int i3rc_hip_find_index(i3rc_hip_integrator *h, int n, const float *table, int64_t m, const float *values, const int32_t *firstGuess,
                        int32_t *out) {
  if (!h) return 1;
  if (n < 1 || m < 1 || !table || !values || !out) return h->fail("i3rc_hip_find_index: bad arguments");
  HIPCHK(h, hipSetDevice(h->device));
  DevBuf dt, dv, dg, dout;
  HIPCHK(h, dt.upload(table, sizeof(float) * (size_t)n)); HIPCHK(h, dv.upload(values, sizeof(float) * (size_t)m));
  if (firstGuess) HIPCHK(h, dg.upload(firstGuess, sizeof(int32_t) * (size_t)m));
  HIPCHK(h, dout.alloc(sizeof(int32_t) * (size_t)m));
  hipLaunchKernelGGL(find_index_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, h->stream, n, (const float *)dt.p, (long long)m,
                     (const float *)dv.p, firstGuess ? (const int32_t *)dg.p : nullptr, (int32_t *)dout.p);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(out, dout.p, sizeof(int32_t) * (size_t)m, hipMemcpyDeviceToHost));
  return 0;
}

/* Test hook: computeSurfaceReflectance (Code/surfaceProperties.f95:121-162) on the device for the surface of i3rc_hip_set_surface. */
int i3rc_hip_surface_reflectance(i3rc_hip_integrator *h, int64_t m, const float *x, const float *y, float *out) {
  if (!h) return 1;
  if (m < 1 || !x || !y || !out) return h->fail("i3rc_hip_surface_reflectance: bad arguments");
  if (!h->dBrdf.p) return h->fail("i3rc_hip_surface_reflectance: no surface description set");
  HIPCHK(h, hipSetDevice(h->device));
  DevBuf dx, dy, dout;
  HIPCHK(h, dx.upload(x, sizeof(float) * (size_t)m)); HIPCHK(h, dy.upload(y, sizeof(float) * (size_t)m));
  HIPCHK(h, dout.alloc(sizeof(float) * (size_t)m));
  SurfaceOnly S;
  S.xsE = (const float *)h->dXs.p; S.ysE = (const float *)h->dYs.p; S.brdf = (const float *)h->dBrdf.p; S.nxs = h->nxs; S.nys = h->nys;
  hipLaunchKernelGGL(surface_reflectance_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, h->stream, S, (long long)m,
                     (const float *)dx.p, (const float *)dy.p, (float *)dout.p);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(out, dout.p, sizeof(float) * (size_t)m, hipMemcpyDeviceToHost));
  return 0;
}

int i3rc_hip_synchronize(i3rc_hip_integrator *h) {
  if (!h) return 1;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}

int i3rc_hip_fetch_tallies(i3rc_hip_integrator *h, double *host) {
  if (!h) return 1;
  if (!host) return h->fail("i3rc_hip_fetch_tallies: null buffer");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(host, h->tally, (size_t)h->layout.total * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

int i3rc_hip_kernel_ms_history(i3rc_hip_integrator *h, int n, float *ms) {
  if (!h || !ms || n < 1) return 1;
  if (n > i3rc_hip_integrator::kEventRing || n > h->timedLaunches)
    return h->fail("i3rc_hip_kernel_ms_history: fewer timed launches recorded than requested (ring of 64)");
  HIPCHK(h, hipSetDevice(h->device));
  for (int k = 0; k < n; ++k) {  // ms[0] = oldest of the last n launches
    const int slot = (int)((h->timedLaunches - n + k) % i3rc_hip_integrator::kEventRing);
    HIPCHK(h, hipEventSynchronize(h->evStop[slot]));
    HIPCHK(h, hipEventElapsedTime(&ms[k], h->evStart[slot], h->evStop[slot]));
  }
  return 0;
}

int i3rc_hip_last_kernel_ms(i3rc_hip_integrator *h, float *ms) { return i3rc_hip_kernel_ms_history(h, 1, ms); }

int64_t i3rc_hip_timed_launch_count(const i3rc_hip_integrator *h) { return h ? (int64_t)h->timedLaunches : 0; }

const char *i3rc_hip_last_kernel_name(const i3rc_hip_integrator *h) { return h ? h->lastKernelName.c_str() : ""; }

int i3rc_hip_normalise(const i3rc_hip_integrator *h, const double *t, float *fluxUp, float *fluxDown, float *fluxAbsorbed,
                       float *volumeAbsorption, float *intensity, float *intensityByComponent) {
  if (!h || !t) return 1;
  const i3rc_tally_layout &L = h->layout;
  const size_t ncol = (size_t)h->nx * h->ny;
  const int nDir = h->nDir, ncomp = h->ncomp;
  const double nPhot = t[L.counters + I3RC_CNT_PHOTONS];
  // photons per column :353-367 (float64 here; the reference works in real(4))
  std::vector<double> perCol(ncol);
  if (h->xyRegular) {
    for (size_t k = 0; k < ncol; ++k) perCol[k] = nPhot / (double)ncol;
  } else {
    const double ax = (double)h->xE.back() - h->xE.front(), ay = (double)h->yE.back() - h->yE.front();
    for (int j = 0; j < h->ny; ++j)
      for (int i = 0; i < h->nx; ++i)
        perCol[(size_t)j * h->nx + i] = (((double)h->yE[j + 1] - h->yE[j]) * ((double)h->xE[i + 1] - h->xE[i])) / (ax * ay) * nPhot;
  }
  for (size_t k = 0; k < ncol; ++k) {
    if (fluxUp) fluxUp[k] = (float)(t[L.fluxUp + k] / perCol[k]);
    if (fluxDown) fluxDown[k] = (float)(t[L.fluxDown + k] / perCol[k]);
    if (fluxAbsorbed) fluxAbsorbed[k] = (float)(t[L.fluxAbsorbed + k] / perCol[k]);
  }
  if (volumeAbsorption)
    for (int kz = 0; kz < h->nz; ++kz)
      for (size_t k = 0; k < ncol; ++k)
        volumeAbsorption[(size_t)kz * ncol + k] =
            (float)(t[L.volumeAbsorption + (size_t)kz * ncol + k] / (perCol[k] * ((double)h->zE[kz + 1] - h->zE[kz])));
  if (nDir > 0 && (intensity || intensityByComponent)) {
    // intensity = sum over components (0 = surface) of intensityByComponent :574-579,:662-667
    std::vector<double> byc((size_t)(ncomp + 1) * nDir * ncol), tot((size_t)nDir * ncol, 0.0);
    for (size_t i = 0; i < byc.size(); ++i) byc[i] = t[L.intensityByComponent + i];
    for (int j = 0; j <= ncomp; ++j)
      for (size_t i = 0; i < (size_t)nDir * ncol; ++i) tot[i] += byc[(size_t)j * nDir * ncol + i];
    if (h->params.limitIntensityContributions) {  // :327-347
      for (int j = 0; j <= ncomp; ++j)
        for (int d = 0; d < nDir; ++d) {
          const double ex = t[L.intensityExcess + (size_t)j * nDir + d];
          if (ex > 0.0) {
            double *b = byc.data() + ((size_t)j * nDir + d) * ncol;
            double s = 0.0;
            for (size_t k = 0; k < ncol; ++k) s += b[k];
            for (size_t k = 0; k < ncol; ++k) {
              const double add = (b[k] / s) * ex;
              tot[(size_t)d * ncol + k] += add;
              b[k] += add;
            }
          }
        }
    }
    if (intensity)
      for (int d = 0; d < nDir; ++d)
        for (size_t k = 0; k < ncol; ++k) intensity[(size_t)d * ncol + k] = (float)(tot[(size_t)d * ncol + k] / perCol[k]);
    if (intensityByComponent)
      for (int j = 0; j <= ncomp; ++j)
        for (int d = 0; d < nDir; ++d)
          for (size_t k = 0; k < ncol; ++k) {
            const size_t o = ((size_t)j * nDir + d) * ncol + k;
            // the reference leaves component 0 un-normalised (:390 loops j = 1:numComponents)
            intensityByComponent[o] = (float)(j == 0 ? byc[o] : byc[o] / perCol[k]);
          }
  }
  return 0;
}

}  // extern "C"
