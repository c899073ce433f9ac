// Microbenchmark: ONE hop of the sharded walk between two PROCESSES through DEVICE memory -- each process owns an inbox
// in its GPU's memory, the other maps it (hipIpcGetMemHandle / hipIpcOpenMemHandle) and stores into it; the owner polls
// its OWN memory (csrc/comm.hip, ring_setup_device).  On a node: two GPUs, the store crosses xGMI; on the
// one-GPU box both processes share the device (same-device IPC) and the figure is the hop without the link.
// Variants of the inbox's memory: hipMalloc (coarse-grained), hipExtMallocWithFlags fine-grained / uncached.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/ipc_hop.cpp -o tools/micro/ipc_hop.bin && tools/micro/ipc_hop.bin
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("[%d] %s: %s\n", (int)getpid(), #x, hipGetErrorString(e_)); fflush(stdout); _exit(1); } } while (0)

__global__ void side(unsigned long long *mine, unsigned long long *theirs, int hops, int first, long long *ticks, int *fail) {
  if (threadIdx.x != 0) return;
  long long t0 = 0;
  for (int h = 1; h <= hops; h++) {
    if (first) {
      if (h == 1) t0 = wall_clock64();
      __hip_atomic_store(theirs, (unsigned long long)h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    long long spins = 0;
    while (true) {
      const unsigned long long v = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (v == (unsigned long long)h) break;
      if (++spins > 20000000ll) {  // (an exit every wave reaches: ~10 s)
        *fail = h;
        return;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    if (!first) __hip_atomic_store(theirs, (unsigned long long)h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (first) *ticks = wall_clock64() - t0;
}

static int rd(int fd, void *p, size_t n) { return read(fd, p, n) == (ssize_t)n ? 0 : 1; }
static int wr(int fd, const void *p, size_t n) { return write(fd, p, n) == (ssize_t)n ? 0 : 1; }

static int run_side(int me, int to_peer, int from_peer, int device) {
  CHECK(hipSetDevice(device));
  const char *names[3] = {"hipMalloc (coarse-grained)", "hipExtMallocWithFlags fine-grained", "hipExtMallocWithFlags uncached"};
  for (int variant = 0; variant < 3; variant++) {
    unsigned long long *inbox = nullptr;
    hipError_t e = variant == 0 ? hipMalloc((void **)&inbox, 4096)
                                : hipExtMallocWithFlags((void **)&inbox, 4096, variant == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached);
    int ok = e == hipSuccess;
    hipIpcMemHandle_t mine_h, peer_h;
    memset(&mine_h, 0, sizeof mine_h);
    if (ok) {
      CHECK(hipMemset(inbox, 0, 4096));
      CHECK(hipDeviceSynchronize());
      ok = hipIpcGetMemHandle(&mine_h, inbox) == hipSuccess;
    }
    (void)hipGetLastError();
    int peer_ok = 0;
    if (wr(to_peer, &ok, sizeof ok) || wr(to_peer, &mine_h, sizeof mine_h) || rd(from_peer, &peer_ok, sizeof peer_ok) || rd(from_peer, &peer_h, sizeof peer_h)) return 1;
    unsigned long long *theirs = nullptr;
    int opened = 0;
    if (ok && peer_ok) opened = hipIpcOpenMemHandle((void **)&theirs, peer_h, hipIpcMemLazyEnablePeerAccess) == hipSuccess;
    (void)hipGetLastError();
    int peer_opened = 0;
    if (wr(to_peer, &opened, sizeof opened) || rd(from_peer, &peer_opened, sizeof peer_opened)) return 1;
    if (!(opened && peer_opened)) {
      if (me == 0) printf("%-40s: not available (alloc/handle %d/%d, open %d/%d)\n", names[variant], ok, peer_ok, opened, peer_opened);
      if (opened) (void)hipIpcCloseMemHandle(theirs);
      if (inbox) (void)hipFree(inbox);
      continue;
    }
    long long *d_ticks;
    int *d_fail;
    CHECK(hipMalloc((void **)&d_ticks, 8));
    CHECK(hipMalloc((void **)&d_fail, 4));
    CHECK(hipMemset(d_ticks, 0, 8));
    CHECK(hipMemset(d_fail, 0, 4));
    CHECK(hipDeviceSynchronize());
    const int hops = 2000;
    hipLaunchKernelGGL(side, dim3(1), dim3(64), 0, 0, inbox, theirs, hops, me == 0 ? 1 : 0, d_ticks, d_fail);
    CHECK(hipDeviceSynchronize());
    long long t = 0;
    int f = 0;
    CHECK(hipMemcpy(&t, d_ticks, 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&f, d_fail, 4, hipMemcpyDeviceToHost));
    int peer_f = 0;
    if (wr(to_peer, &f, sizeof f) || rd(from_peer, &peer_f, sizeof peer_f)) return 1;
    if (me == 0) {
      if (f || peer_f) printf("%-40s: STALE -- a poll never saw hop %d (the owner's loads do not see the peer's stores)\n", names[variant], f ? f : peer_f);
      else printf("%-40s: %.2f us per hop (one way), two processes, devices %d\n", names[variant], (double)t / 100.0 / (2.0 * hops), device);
      fflush(stdout);
    }
    CHECK(hipIpcCloseMemHandle(theirs));
    // (the peer may still have this inbox open: both sides are past their kernels -- the exchange above -- before either frees)
    int done = 1, peer_done = 0;
    if (wr(to_peer, &done, sizeof done) || rd(from_peer, &peer_done, sizeof peer_done)) return 1;
    CHECK(hipFree(inbox));
    CHECK(hipFree(d_ticks));
    CHECK(hipFree(d_fail));
  }
  return 0;
}

int main(int argc, char **argv) {
  int ab[2], ba[2];
  if (pipe(ab) || pipe(ba)) return 1;
  const int dev_b = argc > 1 ? atoi(argv[1]) : 0;  // (a second GPU, where there is one: `ipc_hop.bin 1`)
  const pid_t pid = fork();  // (before anything touches the GPU)
  if (pid == 0) {
    close(ab[1]);
    close(ba[0]);
    _exit(run_side(1, ba[1], ab[0], dev_b));
  }
  close(ab[0]);
  close(ba[1]);
  const int rc = run_side(0, ab[1], ba[0], 0);
  int st = 0;
  waitpid(pid, &st, 0);
  return rc || st;
}
