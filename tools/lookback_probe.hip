// lookback_probe.hip -- the experiment behind DESIGN.md's "no inter-workgroup look-back in the radix passes":
// what a chained scan (single-pass radix: tile status = flag + count per digit, decoupled look-back) costs on gfx950
// next to the histogram + spine launches it would replace.
//
// The probe runs the SKELETON of one radix pass over n 32-bit keys, tiles of 4096 keys, 1024-thread workgroups (two per
// CU), digits of 8 bits -- load the tile, count its digits in LDS, [find the tile's bases], store the tile -- three ways:
//   copy        no bases at all (the floor: what the loads, the LDS histogram and the stores cost)
//   lookback    bases by a chained scan: the tile (taken by ticket, so that every earlier tile is resident or done)
//               publishes its 256 digit counts as aggregate words (one relaxed agent-scope store each: flag and count
//               in ONE naturally aligned word, MI355X_MICROARCH.md "granule"), looks back over the rows of the tiles
//               before it -- LOOK rows requested per step, every lane one digit, relaxed agent-scope (sc1) loads -- until
//               it meets an inclusive row, then publishes its own inclusive row
//   histogram   the three-launch form of csrc/radix_sort.h in its cheapest shape: a separate launch that reads the keys
//               and counts the digits of every tile (the time of that launch is what the look-back has to beat)
// and prints the time of each, the average number of rows a tile had to read, and the slowest spin.  A spin that does
// not end within SPIN_LIMIT polls raises a flag and gives up (the probe must never hang the GPU).
//
//   hipcc -O3 --offload-arch=gfx950 -o build/lookback_probe tools/lookback_probe.hip && build/lookback_probe [n_keys]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef uint32_t u32;
#define TILE 4096
#define THREADS 1024
#define BINS 256
#define SPIN_LIMIT (1u << 22)
#ifndef LOOK
#define LOOK 8                 // rows requested per step of the look-back
#endif
#define FLAG_AGG 0x40000000u
#define FLAG_INC 0x80000000u
#define COUNT_MASK 0x3FFFFFFFu

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ u32 ld_agent(const u32 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(u32 *p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// mode 0: copy, 1: look-back
template <int MODE>
__global__ __launch_bounds__(THREADS) void pass_kernel(const u32 *__restrict__ in, u32 *__restrict__ out, u32 n_tiles, int shift,
                                                      u32 *__restrict__ ticket, u32 *__restrict__ status,
                                                      unsigned long long *__restrict__ rows_read, u32 *__restrict__ max_spin,
                                                      u32 *__restrict__ gave_up)
{
    __shared__ u32 hist[BINS];
    __shared__ u32 base[BINS];
    __shared__ u32 s_tile;
    const u32 tid = threadIdx.x;
    if (tid == 0) s_tile = MODE == 1 ? atomicAdd(ticket, 1u) : blockIdx.x;
    if (tid < BINS) hist[tid] = 0;
    __syncthreads();
    const u32 tile = s_tile;
    if (tile >= n_tiles) return;
    const uint4 k = reinterpret_cast<const uint4 *>(in + (size_t)tile * TILE)[tid];
    atomicAdd(&hist[(k.x >> shift) & 255u], 1u);
    atomicAdd(&hist[(k.y >> shift) & 255u], 1u);
    atomicAdd(&hist[(k.z >> shift) & 255u], 1u);
    atomicAdd(&hist[(k.w >> shift) & 255u], 1u);
    __syncthreads();
    u32 add = 0;
    if (MODE == 1) {
        u32 *row = status + (size_t)tile * BINS;
        if (tid < BINS) {                                 // (four wavefronts look back, one digit per lane; the others wait)
            const u32 d = tid, mine = hist[d];
            st_agent(row + d, FLAG_AGG | mine);
            u32 sum = 0, spins = 0, read = 0;
            bool done = false;
            for (int j = (int)tile - 1; j >= 0 && !done; j -= LOOK) {
                u32 v[LOOK];
#pragma unroll
                for (int q = 0; q < LOOK; q++) v[q] = j - q >= 0 ? ld_agent(status + (size_t)(j - q) * BINS + d) : FLAG_INC;
#pragma unroll
                for (int q = 0; q < LOOK; q++) {
                    if (done) break;
                    u32 polls = 0;
                    while (!(v[q] & (FLAG_AGG | FLAG_INC))) {       // not published yet: poll
                        if (++polls > SPIN_LIMIT) { atomicOr(gave_up, 1u); v[q] = FLAG_INC; break; }
                        __builtin_amdgcn_s_sleep(1);
                        v[q] = ld_agent(status + (size_t)(j - q) * BINS + d);
                    }
                    spins = polls > spins ? polls : spins;
                    read++;
                    sum += v[q] & COUNT_MASK;
                    if (v[q] & FLAG_INC) done = true;
                }
            }
            st_agent(row + d, FLAG_INC | ((sum + mine) & COUNT_MASK));
            base[d] = sum;
            if (rows_read) {
                atomicAdd(rows_read, (unsigned long long)read);
                atomicMax(max_spin, spins);
            }
        }
        __syncthreads();
        add = base[(k.x >> shift) & 255u] & 1u;           // (keeps the bases live; the stores below stay coalesced)
    }
    uint4 o = k;
    o.x += add;
    reinterpret_cast<uint4 *>(out + (size_t)tile * TILE)[tid] = o;
}

__global__ __launch_bounds__(256) void hist_kernel(const u32 *__restrict__ in, u32 n_tiles, int shift, u32 *__restrict__ hist_out)
{
    __shared__ u32 bins[4][BINS];
    const u32 w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (int i = 0; i < 4; i++) bins[i][threadIdx.x] = 0;
    __syncthreads();
    const u32 tile = blockIdx.x * 4 + w;
    if (tile < n_tiles) {
        const uint4 *p = reinterpret_cast<const uint4 *>(in + (size_t)tile * TILE) + lane;
        for (int j = 0; j < TILE / 256; j++) {
            const uint4 k = p[j * 64];
            atomicAdd(&bins[w][(k.x >> shift) & 255u], 1u);
            atomicAdd(&bins[w][(k.y >> shift) & 255u], 1u);
            atomicAdd(&bins[w][(k.z >> shift) & 255u], 1u);
            atomicAdd(&bins[w][(k.w >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    for (int i = 0; i < 4; i++)
        if (blockIdx.x * 4 + i < n_tiles) hist_out[(size_t)(blockIdx.x * 4 + i) * BINS + threadIdx.x] = bins[i][threadIdx.x];
}

int main(int argc, char **argv)
{
    const size_t n = argc > 1 ? (size_t)atoll(argv[1]) : (size_t)61142967;
    const u32 n_tiles = (u32)(n / TILE);
    u32 *in, *out, *status, *ticket, *max_spin, *gave_up, *hist;
    unsigned long long *rows;
    CHECK(hipMalloc(&in, (size_t)n_tiles * TILE * 4));
    CHECK(hipMalloc(&out, (size_t)n_tiles * TILE * 4));
    CHECK(hipMalloc(&status, (size_t)n_tiles * BINS * 4));
    CHECK(hipMalloc(&hist, (size_t)n_tiles * BINS * 4));
    CHECK(hipMalloc(&ticket, 64));
    CHECK(hipMalloc(&rows, 8));
    max_spin = ticket + 1; gave_up = ticket + 2;
    std::vector<u32> h((size_t)n_tiles * TILE);
    u32 x = 12345;
    for (auto &v : h) { x = x * 1664525u + 1013904223u; v = x; }
    CHECK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int reps = 10;
    float ms_copy = 0, ms_look = 0, ms_hist = 0;
    unsigned long long total_rows = 0;
    u32 spin = 0, gave = 0;
    for (int r = 0; r < reps + 2; r++) {
        float ms;
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(pass_kernel<0>, dim3(n_tiles), dim3(THREADS), 0, 0, in, out, n_tiles, 8, ticket, status, rows, max_spin, gave_up);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) ms_copy += ms;
        CHECK(hipMemset(status, 0, (size_t)n_tiles * BINS * 4));
        CHECK(hipMemset(ticket, 0, 64));
        CHECK(hipMemset(rows, 0, 8));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(pass_kernel<1>, dim3(n_tiles), dim3(THREADS), 0, 0, in, out, n_tiles, 8, ticket, status, rows, max_spin, gave_up);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) {
            ms_look += ms;
            unsigned long long rr; u32 t[3];
            CHECK(hipMemcpy(&rr, rows, 8, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(t, ticket, 12, hipMemcpyDeviceToHost));
            total_rows += rr; spin = t[1] > spin ? t[1] : spin; gave |= t[2];
        }
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(hist_kernel, dim3((n_tiles + 3) / 4), dim3(256), 0, 0, in, n_tiles, 8, hist);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) ms_hist += ms;
    }
    {   // the last tile's inclusive row must be the digit histogram of all keys
        std::vector<u32> last(BINS), want(BINS, 0);
        CHECK(hipMemcpy(last.data(), status + (size_t)(n_tiles - 1) * BINS, BINS * 4, hipMemcpyDeviceToHost));
        for (auto v : h) want[(v >> 8) & 255u]++;
        bool ok = true;
        for (int d = 0; d < BINS; d++) ok = ok && (last[d] & COUNT_MASK) == want[d] && (last[d] & FLAG_INC);
        printf("chained scan result %s\n", ok ? "correct" : "WRONG");
    }
    printf("keys %zu, tiles %u (4096 keys, 1024 threads)\n", (size_t)n_tiles * TILE, n_tiles);
    printf("pass skeleton without bases   %.3f ms\n", ms_copy / reps);
    printf("pass skeleton with look-back  %.3f ms   (+%.3f)\n", ms_look / reps, (ms_look - ms_copy) / reps);
    printf("separate histogram launch     %.3f ms   <- what the look-back has to beat (plus ~0.012 ms of spine)\n", ms_hist / reps);
    printf("status words read per tile    %.1f lanes-rows (x 4 B; a row is 256 words) = %.1f rows of 1 KiB\n",
           (double)total_rows / reps / n_tiles, (double)total_rows / reps / n_tiles / 256.0);
    printf("longest wait for a row        %u polls%s\n", spin, gave ? "   (A SPIN GAVE UP)" : "");
    return 0;
}
