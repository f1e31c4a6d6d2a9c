// Process layer (include/i3rc_comm.h): one process per GPU, sums of host arrays over processes.
//   backend rccl : ncclAllReduce(float, sum) over xGMI
//   backend shm  : POSIX shared memory + sense-reversing barrier (single node, CPU only; tests)
// Bootstrap of either: rank 0 listens on MASTER_ADDR:MASTER_PORT (I3RC_COMM_PORT overrides the port) and sends every
// other rank a small blob -- the ncclUniqueId, or the name of the shared-memory segment it has just created.  A TCP
// rendezvous leaves nothing behind that a later run could mistake for its own (a file or a named segment would).
#include "../../include/i3rc_comm.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <arpa/inet.h>
#include <netdb.h>
#include <fcntl.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/mman.h>
#include <sys/socket.h>
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

// ---- rendezvous: rank 0 -> everyone, n bytes -----------------------------------------------------------------------
// Its own port: I3RC_COMM_PORT, else MASTER_PORT + 1 -- under torchrun / SLURM launchers MASTER_PORT itself is held by
// the launcher's store, which would answer a connecting rank with bytes that are not ours (hence also the headers below).
// A launcher only promises that MASTER_PORT is free: where MASTER_PORT + 1 may belong to somebody else, set I3RC_COMM_PORT
// (INTEGRATION.md); rank 0 fails at once when it cannot bind the port, the others when the listener there is not ours.
int rendezvous_port() {
  if (std::getenv("I3RC_COMM_PORT")) return std::atoi(std::getenv("I3RC_COMM_PORT"));
  return env_int("MASTER_PORT", nullptr, 29499) + 1;
}
struct RendezvousHeader { uint32_t magic, bytes; };   // rank 0 -> rank r, followed by the blob
struct RendezvousHello { uint32_t magic, rank; };     // rank r -> rank 0, first thing on the connection
constexpr uint32_t kMagic = 0x69337263u;   // "i3rc"

bool send_all(int fd, const void *buf, size_t n) {
  const char *p = (const char *)buf;
  while (n > 0) {
    const ssize_t k = ::send(fd, p, n, MSG_NOSIGNAL);
    if (k <= 0) return false;
    p += k; n -= (size_t)k;
  }
  return true;
}
bool recv_all(int fd, void *buf, size_t n) {
  char *p = (char *)buf;
  while (n > 0) {
    const ssize_t k = ::recv(fd, p, n, 0);
    if (k <= 0) return false;
    p += k; n -= (size_t)k;
  }
  return true;
}

int rendezvous_broadcast(void *blob, size_t n) {
  const int port = rendezvous_port();
  if (g_rank == 0) {
    const int srv = ::socket(AF_INET, SOCK_STREAM, 0);
    if (srv < 0) return fail("i3rc_comm_init: socket() failed");
    int one = 1;
    (void)setsockopt(srv, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    sockaddr_in addr{};
    addr.sin_family = AF_INET; addr.sin_addr.s_addr = htonl(INADDR_ANY); addr.sin_port = htons((uint16_t)port);
    if (::bind(srv, (sockaddr *)&addr, sizeof(addr)) != 0 || ::listen(srv, g_size) != 0) {
      ::close(srv);
      return fail("i3rc_comm_init: cannot listen on port " + std::to_string(port) + " (MASTER_PORT / I3RC_COMM_PORT in use?)");
    }
    timeval tv{120, 0};                                   // a rank that never shows up must not hang the others for ever
    (void)setsockopt(srv, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
    // Every rank says hello first -- the magic word and its rank number -- and only such a connection counts as a peer and
    // is handed the blob: a stray connection (a port scanner, another job probing MASTER_PORT + 1) is closed and forgotten
    // instead of taking a real rank's place.
    std::string seen((size_t)g_size, 0);
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(120);
    for (int peers = 1; peers < g_size;) {
      if (std::chrono::steady_clock::now() > deadline) { ::close(srv); return fail("i3rc_comm_init: timed out waiting for the other processes"); }
      const int fd = ::accept(srv, nullptr, nullptr);
      if (fd < 0) { ::close(srv); return fail("i3rc_comm_init: timed out waiting for the other processes"); }
      timeval hv{5, 0};                                   // a connection that says nothing is not one of ours
      (void)setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &hv, sizeof(hv));
      RendezvousHello hello{0u, 0u};
      if (!recv_all(fd, &hello, sizeof(hello)) || hello.magic != kMagic || hello.rank == 0u || hello.rank >= (uint32_t)g_size ||
          seen[hello.rank]) {
        ::close(fd);
        continue;
      }
      const RendezvousHeader hdr{kMagic, (uint32_t)n};
      const bool ok = send_all(fd, &hdr, sizeof(hdr)) && send_all(fd, blob, n);
      ::close(fd);
      if (!ok) { ::close(srv); return fail("i3rc_comm_init: sending the rendezvous blob failed"); }
      seen[hello.rank] = 1;
      ++peers;
    }
    ::close(srv);
    return 0;
  }
  // MASTER_ADDR may be a host name (torchrun, SLURM): resolve it; IPv4 (rank 0 listens on INADDR_ANY)
  const std::string host = env_str("MASTER_ADDR", "127.0.0.1");
  addrinfo hints{}, *res = nullptr;
  hints.ai_family = AF_INET; hints.ai_socktype = SOCK_STREAM;
  if (getaddrinfo(host.c_str(), std::to_string(port).c_str(), &hints, &res) != 0 || !res)
    return fail("i3rc_comm_init: cannot resolve MASTER_ADDR " + host);
  sockaddr_in addr = *(sockaddr_in *)res->ai_addr;
  freeaddrinfo(res);
  for (int tries = 0; tries < 1200; ++tries) {             // up to 2 minutes for rank 0 to come up
    const int fd = ::socket(AF_INET, SOCK_STREAM, 0);
    if (fd < 0) return fail("i3rc_comm_init: socket() failed");
    if (::connect(fd, (sockaddr *)&addr, sizeof(addr)) == 0) {
      timeval tv{30, 0};                                    // a listener that is not ours may never send anything
      (void)setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
      RendezvousHeader hdr{0u, 0u};
      const RendezvousHello hello{kMagic, (uint32_t)g_rank};
      const bool ok = send_all(fd, &hello, sizeof(hello)) && recv_all(fd, &hdr, sizeof(hdr)) && hdr.magic == kMagic && hdr.bytes == (uint32_t)n && recv_all(fd, blob, n);
      ::close(fd);
      if (ok) return 0;
      return fail("i3rc_comm_init: " + host + ":" + std::to_string(port) + " did not answer with the rendezvous blob "
                  "(another service on that port?  set I3RC_COMM_PORT)");
    }
    ::close(fd);
    std::this_thread::sleep_for(std::chrono::milliseconds(100));
  }
  return fail("i3rc_comm_init: timed out connecting to " + host + ":" + std::to_string(port));
}

// ---- rccl ---------------------------------------------------------------------------------------------------------
ncclComm_t g_comm = nullptr;
hipStream_t g_stream = nullptr;
void *g_dev = nullptr;     // staging buffer of the all-reduce (float or double values)
size_t g_devCap = 0;

int rccl_init() {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail("i3rc_comm_init: no HIP device for the rccl backend");
  if (g_local >= ndev) return fail("i3rc_comm_init: LOCAL_RANK exceeds the number of GPUs of this node");
  if (hipSetDevice(g_local) != hipSuccess) return fail("i3rc_comm_init: hipSetDevice failed");
  ncclUniqueId id;
  std::memset(&id, 0, sizeof(id));
  if (g_rank == 0 && ncclGetUniqueId(&id) != ncclSuccess) return fail("i3rc_comm_init: ncclGetUniqueId failed");
  if (rendezvous_broadcast(&id, sizeof(id))) return 1;
  if (ncclCommInitRank(&g_comm, g_size, id, g_rank) != ncclSuccess) return fail("i3rc_comm_init: ncclCommInitRank failed");
  if (hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return fail("i3rc_comm_init: stream");
  return 0;
}

template <class T>
int rccl_sum(T *v, int64_t n) {
  const size_t bytes = sizeof(T) * (size_t)n;
  if (hipSetDevice(g_local) != hipSuccess) return fail("i3rc_comm_sum: hipSetDevice failed");
  if (bytes > g_devCap) {
    if (g_dev) (void)hipFree(g_dev);
    if (hipMalloc((void **)&g_dev, bytes) != hipSuccess) { g_dev = nullptr; g_devCap = 0; return fail("i3rc_comm_sum: hipMalloc failed"); }
    g_devCap = bytes;
  }
  if (hipMemcpyAsync(g_dev, v, bytes, hipMemcpyHostToDevice, g_stream) != hipSuccess) return fail("i3rc_comm_sum: H2D failed");
  if (ncclAllReduce(g_dev, g_dev, (size_t)n, sizeof(T) == 8 ? ncclDouble : ncclFloat, ncclSum, g_comm, g_stream) != ncclSuccess)
    return fail("i3rc_comm_sum: ncclAllReduce failed");
  if (hipMemcpyAsync(v, g_dev, bytes, hipMemcpyDeviceToHost, g_stream) != hipSuccess) return fail("i3rc_comm_sum: D2H failed");
  if (hipStreamSynchronize(g_stream) != hipSuccess) return fail("i3rc_comm_sum: stream synchronisation failed");
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
  g_shmBytes = 4096 + sizeof(float) * (size_t)kSlotFloats * g_size;
  char name[64] = {0};
  int fd = -1;
  if (g_rank == 0) {   // a fresh segment per run, named after this process; the others learn the name at the rendezvous
    std::snprintf(name, sizeof(name), "/i3rc_comm_%d_%ld", rendezvous_port(), (long)getpid());
    shm_unlink(name);
    fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)g_shmBytes) != 0) return fail(std::string("i3rc_comm_init: cannot create shared memory ") + name);
  }
  if (rendezvous_broadcast(name, sizeof(name))) return 1;
  g_shmName = name;
  if (g_rank != 0) {
    fd = shm_open(name, O_RDWR, 0600);
    if (fd < 0) return fail(std::string("i3rc_comm_init: cannot open shared memory ") + name);
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

template <class T>
int shm_sum(T *v, int64_t n) {
  const int64_t slot = kSlotFloats * (int64_t)sizeof(float) / (int64_t)sizeof(T);   // elements of T per rank's staging slot
  T *const slots = (T *)g_slots;
  for (int64_t off = 0; off < n; off += slot) {
    const int64_t m = std::min<int64_t>(slot, n - off);
    std::memcpy(slots + (size_t)g_rank * slot, v + off, sizeof(T) * (size_t)m);
    if (shm_barrier()) return 1;
    for (int64_t i = 0; i < m; ++i) {   // ranks summed in rank order: every process gets the same result
      T s = 0;
      for (int r = 0; r < g_size; ++r) s += slots[(size_t)r * slot + i];
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
    if (g_size == 1 && env_str("I3RC_COMM_BACKEND", "") != "rccl") {
      g_backend = NONE;   // (a one-rank RCCL communicator on explicit request only: the GPU test of this backend)
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
  if (n <= 0 || g_backend == NONE || !g_ready) return 0;
  if (!values) return fail("i3rc_comm_sum_float: null buffer");
  return g_backend == RCCL ? rccl_sum(values, n) : shm_sum(values, n);
}

int i3rc_comm_sum_double(double *values, int64_t n) {
  if (n <= 0 || g_backend == NONE || !g_ready) return 0;
  if (!values) return fail("i3rc_comm_sum_double: null buffer");
  return g_backend == RCCL ? rccl_sum(values, n) : shm_sum(values, n);
}

int i3rc_comm_barrier(void) {
  if (g_backend == NONE || !g_ready) return 0;
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
