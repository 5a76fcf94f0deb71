// fake_rccl.cpp -- TEST DOUBLE for librccl, never shipped and never loaded by the product unless VDN_RCCL_LIB names it.
//
// A one-GPU box cannot host two RCCL ranks (RCCL refuses two ranks on one device), so the multi-rank logic of
// libvarden_amd.so (exchange plans, per-peer buffers, agglomerated multigrid levels, where the reductions sit) could not run
// on hardware before the driver's own multi-GPU bench.  This library implements the ten RCCL entry points exchange.hip binds
// (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclSend, ncclRecv, ncclAllReduce, ncclAllGather, ncclGroupStart,
// ncclGroupEnd, ncclGetErrorString) for several PROCESSES that share ONE GPU: messages travel through a memory-mapped file,
// every call is made synchronous (stream sync -> D2H -> mailbox -> H2D).  The semantics the product relies on are kept:
// send/recv inside a group complete together at ncclGroupEnd without ordering constraints; collectives are entered by all ranks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <time.h>

typedef struct { char internal[128]; } ncclUniqueId;
struct Header { volatile int nranks; volatile int bar_count; volatile int bar_sense; int pad[13]; };
struct Box { volatile long wseq, rseq; volatile long bytes; long pad[5]; };      // one mailbox per ordered (src, dst) pair
struct Comm { int rank, nranks; char *base; size_t map_bytes; size_t maxmsg; int local_sense; char path[160]; };
struct Op { int kind; void *ptr; size_t bytes; int peer; hipStream_t st; bool done; };
static thread_local int g_depth = 0;
static thread_local std::vector<Op> g_ops;
static thread_local Comm *g_group_comm = nullptr;

static size_t max_msg() { const char *e = getenv("FAKE_RCCL_MAXMSG_MB"); return (size_t)(e ? atol(e) : 48) << 20; }
static size_t red_bytes() { return (size_t)16 << 20; }
static Header *hdr(Comm *c) { return (Header *)c->base; }
static Box *box(Comm *c, int src, int dst) { return (Box *)(c->base + 4096 + (size_t)(src * c->nranks + dst) * (4096 + c->maxmsg)); }
static char *box_data(Comm *c, int src, int dst) { return (char *)box(c, src, dst) + 4096; }
static char *red_slot(Comm *c, int r) { return c->base + 4096 + (size_t)c->nranks * c->nranks * (4096 + c->maxmsg) + (size_t)r * red_bytes(); }
static size_t total_bytes(int nranks, size_t maxmsg) { return 4096 + (size_t)nranks * nranks * (4096 + maxmsg) + (size_t)nranks * red_bytes(); }
static void nap() { struct timespec ts = { 0, 20000 }; nanosleep(&ts, nullptr); }
static void die(const char *m) { fprintf(stderr, "fake_rccl: %s\n", m); abort(); }
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "fake_rccl: %s -> %s\n", #x, hipGetErrorString(e_)); abort(); } } while (0)

static void barrier(Comm *c) {
  Header *h = hdr(c);
  c->local_sense = !c->local_sense;
  if (__atomic_add_fetch(&h->bar_count, 1, __ATOMIC_ACQ_REL) == c->nranks) { h->bar_count = 0; __atomic_store_n(&h->bar_sense, c->local_sense, __ATOMIC_RELEASE); }
  else while (__atomic_load_n(&h->bar_sense, __ATOMIC_ACQUIRE) != c->local_sense) nap();
}
static bool try_send(Comm *c, Op &o) {
  Box *b = box(c, c->rank, o.peer);
  if (__atomic_load_n(&b->wseq, __ATOMIC_ACQUIRE) != __atomic_load_n(&b->rseq, __ATOMIC_ACQUIRE)) return false;     // slot still full
  if (o.bytes > c->maxmsg) die("message larger than FAKE_RCCL_MAXMSG_MB");
  HC(hipMemcpyAsync(box_data(c, c->rank, o.peer), o.ptr, o.bytes, hipMemcpyDeviceToHost, o.st)); HC(hipStreamSynchronize(o.st));
  b->bytes = (long)o.bytes;
  __atomic_add_fetch(&b->wseq, 1, __ATOMIC_RELEASE);
  return true;
}
static bool try_recv(Comm *c, Op &o) {
  Box *b = box(c, o.peer, c->rank);
  if (__atomic_load_n(&b->wseq, __ATOMIC_ACQUIRE) == __atomic_load_n(&b->rseq, __ATOMIC_ACQUIRE)) return false;     // nothing there yet
  if ((size_t)b->bytes != o.bytes) { fprintf(stderr, "fake_rccl: rank %d expects %zu bytes from %d, message has %ld\n", c->rank, o.bytes, o.peer, (long)b->bytes); abort(); }
  HC(hipMemcpyAsync(o.ptr, box_data(c, o.peer, c->rank), o.bytes, hipMemcpyHostToDevice, o.st)); HC(hipStreamSynchronize(o.st));
  __atomic_add_fetch(&b->rseq, 1, __ATOMIC_RELEASE);
  return true;
}
static void run_ops(Comm *c, std::vector<Op> &ops) {
  for (auto &o : ops) HC(hipStreamSynchronize(o.st));
  size_t left = ops.size(); long spins = 0;
  while (left) {
    bool progress = false;
    for (auto &o : ops) {                      // sends to one peer and receives from one peer complete in posting order
      if (o.done) continue;
      bool earlier = false;
      for (auto &p : ops) { if (&p == &o) break; if (!p.done && p.kind == o.kind && p.peer == o.peer) { earlier = true; break; } }
      if (earlier) continue;
      if (o.kind == 0 ? try_send(c, o) : try_recv(c, o)) { o.done = true; left--; progress = true; }
    }
    if (!progress) { nap(); if (++spins > 3000000) die("send/recv group made no progress for 60 s (unmatched message?)"); } else spins = 0;
  }
}

extern "C" {
// the handshake varden_amd/csrc/exchange.hip asks of a library named by VDN_RCCL_LIB (with VDN_TESTING=1): "vdntest"
long vdn_test_transport_magic() { return 0x76646e74657374L; }
const char *ncclGetErrorString(int) { return "fake_rccl error"; }
int ncclGetUniqueId(ncclUniqueId *id) {
  memset(id, 0, sizeof *id);
  const char *dir = getenv("FAKE_RCCL_DIR");
  snprintf(id->internal, sizeof id->internal, "%s/vdn_fake_rccl_%d_%ld", dir ? dir : "/tmp", (int)getpid(), (long)time(nullptr));
  int fd = open(id->internal, O_CREAT | O_RDWR | O_TRUNC, 0600);
  if (fd < 0) return 1;
  close(fd);
  return 0;
}
int ncclCommInitRank(Comm **out, int nranks, ncclUniqueId id, int rank) {
  Comm *c = new Comm(); c->rank = rank; c->nranks = nranks; c->maxmsg = max_msg(); c->local_sense = 0;
  snprintf(c->path, sizeof c->path, "%s", id.internal);
  c->map_bytes = total_bytes(nranks, c->maxmsg);
  int fd = -1;
  for (int tries = 0; tries < 600000 && fd < 0; tries++) { fd = open(c->path, O_RDWR); if (fd < 0) nap(); }
  if (fd < 0) return 1;
  if (rank == 0) { if (ftruncate(fd, (off_t)c->map_bytes) != 0) return 1; }
  else { struct stat sb; for (;;) { if (fstat(fd, &sb) != 0) return 1; if ((size_t)sb.st_size >= c->map_bytes) break; nap(); } }
  c->base = (char *)mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (c->base == (char *)MAP_FAILED) return 1;
  if (rank == 0) __atomic_store_n(&hdr(c)->nranks, nranks, __ATOMIC_RELEASE);
  else while (__atomic_load_n(&hdr(c)->nranks, __ATOMIC_ACQUIRE) != nranks) nap();
  barrier(c);
  *out = c;
  return 0;
}
int ncclCommCount(Comm *c, int *count) { *count = c->nranks; return 0; }
int ncclCommDestroy(Comm *c) {
  barrier(c);
  munmap(c->base, c->map_bytes);
  if (c->rank == 0) unlink(c->path);
  delete c;
  return 0;
}
int ncclGroupStart() { g_depth++; return 0; }
int ncclGroupEnd() {
  if (--g_depth > 0) return 0;
  if (!g_ops.empty()) { run_ops(g_group_comm, g_ops); g_ops.clear(); }
  return 0;
}
static int post(int kind, void *ptr, size_t count, int dtype, int peer, Comm *c, hipStream_t st) {
  if (dtype != 8) die("only ncclFloat64 is implemented");
  Op o; o.kind = kind; o.ptr = ptr; o.bytes = count * 8; o.peer = peer; o.st = st; o.done = false;
  if (g_depth > 0) { g_group_comm = c; g_ops.push_back(o); return 0; }
  std::vector<Op> one(1, o); run_ops(c, one);
  return 0;
}
int ncclSend(const void *buf, size_t count, int dtype, int peer, Comm *c, hipStream_t st) { return post(0, (void *)buf, count, dtype, peer, c, st); }
int ncclRecv(void *buf, size_t count, int dtype, int peer, Comm *c, hipStream_t st) { return post(1, buf, count, dtype, peer, c, st); }
int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, Comm *c, hipStream_t st) {
  if (dtype == 1) {                             // ncclUint8, MAX only (tag bitmaps)
    if (op != 2 || count > red_bytes()) die("allreduce: uint8 MAX up to 16 MB only");
    unsigned char *mine = (unsigned char *)red_slot(c, c->rank);
    HC(hipMemcpyAsync(mine, send, count, hipMemcpyDeviceToHost, st)); HC(hipStreamSynchronize(st));
    barrier(c);
    std::vector<unsigned char> acc(count);
    for (size_t i = 0; i < count; i++) { unsigned char v = 0; for (int r = 0; r < c->nranks; r++) { const unsigned char w = ((unsigned char *)red_slot(c, r))[i]; v = v > w ? v : w; } acc[i] = v; }
    barrier(c);
    HC(hipMemcpyAsync(recv, acc.data(), count, hipMemcpyHostToDevice, st)); HC(hipStreamSynchronize(st));
    return 0;
  }
  if (dtype != 8 || count * 8 > red_bytes()) die("allreduce: f64 up to 16 MB only");
  double *mine = (double *)red_slot(c, c->rank);
  HC(hipMemcpyAsync(mine, send, count * 8, hipMemcpyDeviceToHost, st)); HC(hipStreamSynchronize(st));
  barrier(c);
  std::vector<double> acc(count);
  for (size_t i = 0; i < count; i++) {
    double v = ((double *)red_slot(c, 0))[i];
    for (int r = 1; r < c->nranks; r++) { const double w = ((double *)red_slot(c, r))[i]; if (op == 0) v += w; else if (op == 2) v = v > w ? v : w; else if (op == 3) v = v < w ? v : w; else die("allreduce: op"); }
    acc[i] = v;
  }
  barrier(c);                                   // everybody has read the slots before anyone overwrites them
  HC(hipMemcpyAsync(recv, acc.data(), count * 8, hipMemcpyHostToDevice, st)); HC(hipStreamSynchronize(st));
  return 0;
}
int ncclAllGather(const void *send, void *recv, size_t count, int dtype, Comm *c, hipStream_t st) {
  if (dtype != 8 || count * 8 > red_bytes()) die("allgather: f64 up to 16 MB per rank only");
  HC(hipMemcpyAsync(red_slot(c, c->rank), send, count * 8, hipMemcpyDeviceToHost, st)); HC(hipStreamSynchronize(st));
  barrier(c);
  for (int r = 0; r < c->nranks; r++) HC(hipMemcpyAsync((char *)recv + (size_t)r * count * 8, red_slot(c, r), count * 8, hipMemcpyHostToDevice, st));
  HC(hipStreamSynchronize(st));
  barrier(c);
  return 0;
}
}
