// Probe: cost of one "all-gather + flag barrier" round between co-resident workgroups on MI355X, as a function of
//   * group shape: G groups x NC members, members dealt by blockIdx % G (XCD-aligned under round-robin dispatch)
//     or by blockIdx / NC (contiguous ids -> every group spans all 8 XCDs),
//   * payload store flavour: sc1 write-through (agent-scope safe) vs plain (stays in the XCD's L2),
//   * payload bytes per member.
// Every round each member stores its payload slice, drains, barrier, publishes flag=round; then polls all flags of
// its group, reads the whole group's payload with sc1 loads and folds it into a checksum that is verified on the host.
// Usage: sync_probe   (prints a table)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct P {
  int G, NC, rounds, slice_f4;  // payload per member = slice_f4 * 16 bytes
  int by_mod, plain_store;
  float* buf;        // [2][G][NC*slice_f4] f32x4
  unsigned* flags;   // [G][NC]
  unsigned* status;
  unsigned long long* sums;  // per block
  int* xcc;          // per block
  unsigned long long* cycles;
};

__global__ void __launch_bounds__(256) probe(P p) {
  __shared__ int abort_s;
  const int tid = threadIdx.x, lane = tid & 63;
  const int bid = blockIdx.x;
  const int g = p.by_mod ? bid % p.G : bid / p.NC;
  const int me = p.by_mod ? bid / p.G : bid % p.NC;
  if (tid == 0) {
    abort_s = 0;
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    p.xcc[bid] = (int)(x & 0xf);
  }
  const long grp_f4 = (long)p.NC * p.slice_f4;
  unsigned* flags = p.flags + g * p.NC;
  unsigned long long sum = 0;
  __syncthreads();
  const unsigned long long t0 = wall_clock64();
  for (int r = 1; r <= p.rounds; ++r) {
    float* base = p.buf + ((long)(r & 1) * p.G + g) * grp_f4 * 4;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(grp_f4 * 16), 0x00027000);
    // publish my slice
    for (int i = tid; i < p.slice_f4; i += 256) {
      i32x4 v = {r, me, i, r ^ me};
      const int off = (me * p.slice_f4 + i) * 16;
      if (p.plain_store) __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
      else __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flags + me, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // wait for the group
    if (tid < 64) {
      const unsigned long long w0 = wall_clock64();
      int bad = 0;
      while (true) {
        bool ok = true;
        for (int i = lane; i < p.NC; i += 64) ok = ok && (__hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)r);
        if (__all(ok)) break;
        if (wall_clock64() - w0 > 200000000ull || __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { bad = 1; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      if (lane == 0) { abort_s = bad; if (bad) __hip_atomic_store(p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
    __syncthreads();
    if (abort_s) return;
    // gather everyone's slice
    for (long i = tid; i < grp_f4; i += 256) {
      i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(i * 16), 0, 16);
      sum += (unsigned)v[0] + (unsigned)v[1] * 3u + (unsigned)v[2] * 5u + (unsigned)v[3] * 7u;
    }
  }
  const unsigned long long t1 = wall_clock64();
  // block checksum
  __shared__ unsigned long long red[256];
  red[tid] = sum;
  __syncthreads();
  if (tid == 0) {
    unsigned long long s = 0;
    for (int i = 0; i < 256; ++i) s += red[i];
    p.sums[bid] = s;
    p.cycles[bid] = t1 - t0;
  }
}

static unsigned long long expect(const P& p) {
  unsigned long long s = 0;
  for (int r = 1; r <= p.rounds; ++r)
    for (int me = 0; me < p.NC; ++me)
      for (int i = 0; i < p.slice_f4; ++i) s += (unsigned)r + (unsigned)me * 3u + (unsigned)i * 5u + (unsigned)(r ^ me) * 7u;
  return s;
}

int main() {
  const int rounds = 2000;
  struct Case { int G, NC, by_mod, plain, slice_f4; const char* name; };
  std::vector<Case> cases = {
      {2, 128, 0, 0, 32, "2x128 contiguous ids, sc1 stores, 512 B/member (current LSTM fwd shape)"},
      {2, 128, 1, 0, 32, "2x128 by id%2,         sc1 stores, 512 B/member"},
      {8, 32, 1, 0, 32, "8x32  by id%8 (XCD),   sc1 stores, 512 B/member"},
      {8, 32, 1, 1, 32, "8x32  by id%8 (XCD),   PLAIN stores, 512 B/member"},
      {8, 32, 0, 0, 32, "8x32  contiguous ids,  sc1 stores, 512 B/member"},
      {8, 32, 0, 1, 32, "8x32  contiguous ids,  PLAIN stores (expected WRONG/slow: cross-XCD)"},
      {8, 32, 1, 1, 128, "8x32  by id%8 (XCD),   PLAIN stores, 2 KB/member (bwd dG, 8 rows)"},
      {8, 32, 1, 0, 128, "8x32  by id%8 (XCD),   sc1 stores, 2 KB/member"},
      {2, 128, 0, 0, 128, "2x128 contiguous ids, sc1 stores, 2 KB/member (current LSTM bwd shape)"},
      {8, 32, 1, 1, 8, "8x32  by id%8 (XCD),   PLAIN stores, 128 B/member"},
      {8, 32, 1, 0, 8, "8x32  by id%8 (XCD),   sc1 stores, 128 B/member"},
      {1, 256, 0, 0, 8, "1x256 all,             sc1 stores, 128 B/member"},
  };
  for (auto& c : cases) {
    P p{};
    p.G = c.G; p.NC = c.NC; p.rounds = rounds; p.slice_f4 = c.slice_f4; p.by_mod = c.by_mod; p.plain_store = c.plain;
    const int nb = c.G * c.NC;
    const size_t buf_bytes = (size_t)2 * c.G * c.NC * c.slice_f4 * 16;
    CK(hipMalloc(&p.buf, buf_bytes)); CK(hipMemset(p.buf, 0, buf_bytes));
    CK(hipMalloc(&p.flags, nb * 4 + 64)); CK(hipMemset(p.flags, 0, nb * 4 + 64));
    p.status = p.flags + nb;
    CK(hipMalloc(&p.sums, nb * 8)); CK(hipMalloc(&p.xcc, nb * 4)); CK(hipMalloc(&p.cycles, nb * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 0, 0, p);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> sums(nb), cyc(nb); std::vector<int> xcc(nb); unsigned st = 0;
    CK(hipMemcpy(sums.data(), p.sums, nb * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(xcc.data(), p.xcc, nb * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(cyc.data(), p.cycles, nb * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&st, p.status, 4, hipMemcpyDeviceToHost));
    const unsigned long long want = expect(p);
    int wrong = 0, mixed = 0;
    for (int i = 0; i < nb; ++i) wrong += sums[i] != want;
    for (int g = 0; g < c.G; ++g) {  // does every group sit on one XCD?
      int first = -1;
      for (int i = 0; i < nb; ++i) {
        const int gi = c.by_mod ? i % c.G : i / c.NC;
        if (gi != g) continue;
        if (first < 0) first = xcc[i]; else if (xcc[i] != first) { mixed++; break; }
      }
    }
    printf("%-78s: %7.3f us/round  status=%u wrong_blocks=%d groups_spanning_xcds=%d/%d\n", c.name, 1e3 * ms / rounds, st, wrong, mixed, c.G);
    CK(hipFree(p.buf)); CK(hipFree(p.flags)); CK(hipFree(p.sums)); CK(hipFree(p.xcc)); CK(hipFree(p.cycles));
  }
  return 0;
}
