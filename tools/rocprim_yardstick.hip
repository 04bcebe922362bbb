// rocprim_yardstick.hip -- an EXTERNAL yardstick for the radix sort of the level-0 window keys (measurement only: the
// product never includes, links or calls rocPRIM).
//
// The all-suffix window sort of BASELINE configs[1] sorts 61 142 967 (u32 key, u32 suffix) pairs by 24 key bits in
// three 8-bit passes of three launches each (histogram, spine, scatter; csrc/radix_sort.h) and orders the last 8 bits
// in LDS inside the fused finish.  rocPRIM ships with the image; its device radix sort (onesweep: one histogram launch
// for all digits, then one launch per digit with decoupled look-back) on the same number of pairs, same key width, says
// what a tuned library gets out of this chip for the same job:
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/rocprim_yardstick.hip -o /tmp/rocprim_yardstick && /tmp/rocprim_yardstick
#include <cstring>                           // (rocprim's texture iterator calls memset on the host)
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/rocprim_version.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static double time_sort(const uint32_t *k_in, uint32_t *k_out, const uint32_t *v_in, uint32_t *v_out, size_t n, unsigned begin_bit,
                        unsigned end_bit, void *tmp, size_t tmp_bytes, int reps)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int r = 0; r < reps + 2; r++) {
        CHECK(hipEventRecord(e0));
        CHECK(rocprim::radix_sort_pairs(tmp, tmp_bytes, k_in, k_out, v_in, v_out, n, begin_bit, end_bit));
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float t = 0;
        CHECK(hipEventElapsedTime(&t, e0, e1));
        if (r >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int main(int argc, char **argv)
{
    const size_t n = argc > 1 ? (size_t)atoll(argv[1]) : 61142967;
    std::vector<uint32_t> h(n);
    std::mt19937 rng(20240 + 2);
    // keys shaped like the window keys of random A-Z word text: six 5-bit symbol fields of 27 values + 2 bits
    for (size_t i = 0; i < n; i++) {
        uint32_t k = 0;
        for (int f = 0; f < 6; f++) k = (k << 5) | (rng() % 27 + 1);
        h[i] = (k << 2) | (rng() & 3);
    }
    uint32_t *k_in, *k_out, *v_in, *v_out;
    CHECK(hipMalloc(&k_in, n * 4)); CHECK(hipMalloc(&k_out, n * 4)); CHECK(hipMalloc(&v_in, n * 4)); CHECK(hipMalloc(&v_out, n * 4));
    CHECK(hipMemcpy(k_in, h.data(), n * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < n; i++) h[i] = (uint32_t)i;
    CHECK(hipMemcpy(v_in, h.data(), n * 4, hipMemcpyHostToDevice));
    size_t tmp_bytes = 0;
    CHECK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k_in, k_out, v_in, v_out, n, 0, 32));
    void *tmp;
    CHECK(hipMalloc(&tmp, tmp_bytes));
    printf("rocPRIM %d.%d.%d radix_sort_pairs, %zu (u32, u32) pairs, %.1f MB of temporary storage\n", ROCPRIM_VERSION_MAJOR, ROCPRIM_VERSION_MINOR,
           ROCPRIM_VERSION_PATCH, n, tmp_bytes / 1e6);
    const struct { unsigned b, e; const char *what; } cases[] = {
        {8, 32, "bits 8..32 (24 bits: what the three global passes of the window sort order)"},
        {0, 32, "bits 0..32 (all 32 bits: with the digit the fused finish orders in LDS)"},
        {24, 32, "bits 24..32 (one digit: the price of one pass)"}};
    for (const auto &c : cases) {
        const double ms = time_sort(k_in, k_out, v_in, v_out, n, c.b, c.e, tmp, tmp_bytes, 9);
        const double passes = (c.e - c.b) / 8.0;
        printf("  %-80s %.3f ms  (%.3f ms per 8 bits; 16 B x n per pass = %.0f GB/s)\n", c.what, ms, ms / passes, 16.0 * n * passes / (ms * 1e-3) / 1e9);
    }
    // verify the last sort (bits 24..32) is a stable sort by the top byte
    std::vector<uint32_t> ko(n), vo(n);
    CHECK(hipMemcpy(ko.data(), k_out, n * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(vo.data(), v_out, n * 4, hipMemcpyDeviceToHost));
    for (size_t i = 1; i < n; i++)
        if ((ko[i - 1] >> 24) > (ko[i] >> 24) || ((ko[i - 1] >> 24) == (ko[i] >> 24) && vo[i - 1] > vo[i])) { printf("NOT a stable sort at %zu\n", i); return 1; }
    printf("  (output checked: stable by the sorted bits)\n");
    return 0;
}
