// Process layer (include/i3rc_comm.h): one process per GPU, sums of host arrays over processes.
//   backend rccl : ncclAllReduce(float, sum) over xGMI; bootstrap = ncclUniqueId passed through a file
//   backend shm  : POSIX shared memory + sense-reversing barrier (single node, CPU only; tests)
#include "../../include/i3rc_comm.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

namespace {

std::string g_err;
int g_size = 1, g_rank = 0, g_local = 0;
bool g_ready = false;
enum Backend { NONE, RCCL, SHM } g_backend = NONE;

int fail(const std::string &m) { g_err = m; return 1; }

int env_int(const char *a, const char *b, int dflt) {
  const char *v = std::getenv(a);
  if (!v && b) v = std::getenv(b);
  return v ? std::atoi(v) : dflt;
}
std::string env_str(const char *a, const char *dflt) {
  const char *v = std::getenv(a);
  return v ? v : dflt;
}

// ---- rccl ---------------------------------------------------------------------------------------------------------
ncclComm_t g_comm = nullptr;
hipStream_t g_stream = nullptr;
float *g_dev = nullptr;
size_t g_devCap = 0;

int rccl_init() {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail("i3rc_comm_init: no HIP device for the rccl backend");
  if (g_local >= ndev) return fail("i3rc_comm_init: LOCAL_RANK exceeds the number of GPUs of this node");
  if (hipSetDevice(g_local) != hipSuccess) return fail("i3rc_comm_init: hipSetDevice failed");
  const std::string path = env_str("I3RC_COMM_DIR", "/dev/shm") + "/i3rc_nccl_" + env_str("MASTER_PORT", "29500") + ".id";
  ncclUniqueId id;
  if (g_rank == 0) {
    if (ncclGetUniqueId(&id) != ncclSuccess) return fail("i3rc_comm_init: ncclGetUniqueId failed");
    const std::string tmp = path + ".tmp";
    FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f || std::fwrite(&id, sizeof(id), 1, f) != 1) return fail("i3rc_comm_init: cannot write " + tmp);
    std::fclose(f);
    if (std::rename(tmp.c_str(), path.c_str()) != 0) return fail("i3rc_comm_init: cannot publish " + path);
  } else {
    bool got = false;
    for (int tries = 0; tries < 1200 && !got; ++tries) {   // up to 2 minutes
      FILE *f = std::fopen(path.c_str(), "rb");
      if (f) {
        got = std::fread(&id, sizeof(id), 1, f) == 1;
        std::fclose(f);
      }
      if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    if (!got) return fail("i3rc_comm_init: timed out waiting for " + path);
  }
  if (ncclCommInitRank(&g_comm, g_size, id, g_rank) != ncclSuccess) return fail("i3rc_comm_init: ncclCommInitRank failed");
  if (hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return fail("i3rc_comm_init: stream");
  return 0;
}

int rccl_sum(float *v, int64_t n) {
  const size_t bytes = sizeof(float) * (size_t)n;
  if (hipSetDevice(g_local) != hipSuccess) return fail("i3rc_comm_sum_float: hipSetDevice failed");
  if (bytes > g_devCap) {
    if (g_dev) (void)hipFree(g_dev);
    if (hipMalloc((void **)&g_dev, bytes) != hipSuccess) { g_dev = nullptr; g_devCap = 0; return fail("i3rc_comm_sum_float: hipMalloc failed"); }
    g_devCap = bytes;
  }
  if (hipMemcpyAsync(g_dev, v, bytes, hipMemcpyHostToDevice, g_stream) != hipSuccess) return fail("i3rc_comm_sum_float: H2D failed");
  if (ncclAllReduce(g_dev, g_dev, (size_t)n, ncclFloat, ncclSum, g_comm, g_stream) != ncclSuccess)
    return fail("i3rc_comm_sum_float: ncclAllReduce failed");
  if (hipMemcpyAsync(v, g_dev, bytes, hipMemcpyDeviceToHost, g_stream) != hipSuccess) return fail("i3rc_comm_sum_float: D2H failed");
  if (hipStreamSynchronize(g_stream) != hipSuccess) return fail("i3rc_comm_sum_float: stream synchronisation failed");
  return 0;
}

// ---- shm ----------------------------------------------------------------------------------------------------------
constexpr int64_t kSlotFloats = 1 << 20;   // per-rank staging slot (4 MB); longer arrays go in pieces
struct ShmHeader {
  volatile int arrived;      // barrier counter
  volatile int generation;   // barrier sense
  volatile int attached;     // ranks that mapped the segment
};
ShmHeader *g_hdr = nullptr;
float *g_slots = nullptr;
std::string g_shmName;
size_t g_shmBytes = 0;

int shm_barrier() {
  const int gen = __atomic_load_n(&g_hdr->generation, __ATOMIC_ACQUIRE);
  if (__atomic_add_fetch(&g_hdr->arrived, 1, __ATOMIC_ACQ_REL) == g_size) {
    __atomic_store_n(&g_hdr->arrived, 0, __ATOMIC_RELAXED);
    __atomic_add_fetch(&g_hdr->generation, 1, __ATOMIC_ACQ_REL);
  } else {
    long spins = 0;
    while (__atomic_load_n(&g_hdr->generation, __ATOMIC_ACQUIRE) == gen) {
      if (++spins > 100) std::this_thread::sleep_for(std::chrono::microseconds(50));
      if (spins > 2400000) return fail("i3rc_comm: barrier timed out (a process died?)");   // ~2 minutes
    }
  }
  return 0;
}

int shm_init() {
  g_shmName = "/i3rc_comm_" + env_str("MASTER_PORT", "29500");
  g_shmBytes = 4096 + sizeof(float) * (size_t)kSlotFloats * g_size;
  int fd = -1;
  if (g_rank == 0) {
    shm_unlink(g_shmName.c_str());
    fd = shm_open(g_shmName.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)g_shmBytes) != 0) return fail("i3rc_comm_init: cannot create shared memory " + g_shmName);
  } else {
    for (int tries = 0; tries < 1200 && fd < 0; ++tries) {
      fd = shm_open(g_shmName.c_str(), O_RDWR, 0600);
      struct stat st;
      if (fd >= 0 && (fstat(fd, &st) != 0 || (size_t)st.st_size < g_shmBytes)) { close(fd); fd = -1; }
      if (fd < 0) std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    if (fd < 0) return fail("i3rc_comm_init: timed out waiting for shared memory " + g_shmName);
  }
  void *p = mmap(nullptr, g_shmBytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return fail("i3rc_comm_init: mmap failed");
  g_hdr = (ShmHeader *)p;
  g_slots = (float *)((char *)p + 4096);
  __atomic_add_fetch(&g_hdr->attached, 1, __ATOMIC_ACQ_REL);
  for (long spins = 0; __atomic_load_n(&g_hdr->attached, __ATOMIC_ACQUIRE) < g_size; ++spins) {
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
    if (spins > 120000) return fail("i3rc_comm_init: not all processes attached");
  }
  return 0;
}

int shm_sum(float *v, int64_t n) {
  for (int64_t off = 0; off < n; off += kSlotFloats) {
    const int64_t m = std::min<int64_t>(kSlotFloats, n - off);
    std::memcpy(g_slots + (size_t)g_rank * kSlotFloats, v + off, sizeof(float) * (size_t)m);
    if (shm_barrier()) return 1;
    for (int64_t i = 0; i < m; ++i) {   // ranks summed in rank order: every process gets the same float32 result
      float s = 0.0f;
      for (int r = 0; r < g_size; ++r) s += g_slots[(size_t)r * kSlotFloats + i];
      v[off + i] = s;
    }
    if (shm_barrier()) return 1;
  }
  return 0;
}

}  // namespace

extern "C" {

const char *i3rc_comm_last_error(void) { return g_err.c_str(); }

int i3rc_comm_init(int *numProcs, int *thisProc) {
  if (!numProcs || !thisProc) return fail("i3rc_comm_init: null argument");
  if (!g_ready) {
    g_size = env_int("WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", 1);
    g_rank = env_int("RANK", "OMPI_COMM_WORLD_RANK", 0);
    g_local = env_int("LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", g_rank);
    if (g_size < 1 || g_rank < 0 || g_rank >= g_size) return fail("i3rc_comm_init: inconsistent RANK / WORLD_SIZE");
    if (g_size == 1) {
      g_backend = NONE;
    } else if (env_str("I3RC_COMM_BACKEND", "rccl") == "shm") {
      g_backend = SHM;
      if (shm_init()) return 1;
    } else {
      g_backend = RCCL;
      if (rccl_init()) return 1;
    }
    g_ready = true;
  }
  *numProcs = g_size;
  *thisProc = g_rank;
  return 0;
}

int i3rc_comm_local_device(void) { return g_backend == SHM ? 0 : g_local; }

int i3rc_comm_sum_float(float *values, int64_t n) {
  if (n <= 0 || g_size == 1 || !g_ready) return 0;
  if (!values) return fail("i3rc_comm_sum_float: null buffer");
  return g_backend == RCCL ? rccl_sum(values, n) : shm_sum(values, n);
}

int i3rc_comm_barrier(void) {
  if (g_size == 1 || !g_ready) return 0;
  if (g_backend == SHM) return shm_barrier();
  float one = 1.0f;
  return rccl_sum(&one, 1);
}

int i3rc_comm_finalize(void) {
  if (!g_ready) return 0;
  if (g_backend == RCCL) {
    (void)i3rc_comm_barrier();
    if (g_comm) (void)ncclCommDestroy(g_comm);
    if (g_dev) (void)hipFree(g_dev);
    if (g_stream) (void)hipStreamDestroy(g_stream);
    g_comm = nullptr; g_dev = nullptr; g_stream = nullptr; g_devCap = 0;
    if (g_rank == 0) std::remove((env_str("I3RC_COMM_DIR", "/dev/shm") + "/i3rc_nccl_" + env_str("MASTER_PORT", "29500") + ".id").c_str());
  } else if (g_backend == SHM) {
    (void)shm_barrier();
    munmap((void *)g_hdr, g_shmBytes);
    if (g_rank == 0) shm_unlink(g_shmName.c_str());
    g_hdr = nullptr; g_slots = nullptr;
  }
  g_ready = false;
  g_backend = NONE;
  return 0;
}

}  // extern "C"
