// vmem_microbench.hip — what a vector memory instruction costs a CU on gfx950 when the lanes of a wave read
// DIFFERENT records (the access pattern of the BVH walks: every lane fetches its own node).
//
//   hipcc --offload-arch=gfx950 -O2 tools/vmem_microbench.hip -o gpurun_out/vmem_microbench
//   gpurun_out/vmem_microbench > gpurun_out/vmem_microbench.json
//
// Method: a table of 64-byte records that stays in the L2 (4 MiB by default; also 256 KiB and 64 MiB); every lane walks
// its own pseudo-random sequence of records (an LCG per lane — independent of the loaded data, so the loads of one
// iteration are independent and many are in flight: this measures THROUGHPUT, not latency); one iteration issues the
// pattern under test against one record and folds the result into an accumulator.  256-thread workgroups, exactly
// CUs × W of them (W = 1, 2, 4, 6 waves per SIMD resident together).  Reported from the wall clock × the shader clock:
//   cyc_per_inst_cu = kernel cycles / (vector memory instructions a CU issued)
// Patterns: one dword / dwordx2 / dwordx4 of the record; 2, 3 and 4 dwordx4 of the same record (a 32-, 48-, 64-byte
// node); dwordx4 + a dword from a second table (box + separate link); the same loads with all lanes on ONE record
// (wave-uniform address) and with lane-consecutive addresses (coalesced), for contrast.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

enum { P_DWORD, P_X2, P_X4, P_2X4, P_3X4, P_4X4, P_X4_PLUS_DWORD, P_2X4_PLUS_DWORD, P_COUNT };
static const char *pattern_name[P_COUNT] = {"dword", "dwordx2", "dwordx4", "2 x dwordx4 (32-byte record)", "3 x dwordx4 (48 bytes)",
                                            "4 x dwordx4 (64 bytes)", "dwordx4 + dword of a second table", "2 x dwordx4 + dword of a second table"};
static const int pattern_insts[P_COUNT] = {1, 1, 1, 2, 3, 4, 2, 3};

// ADDR 0: every lane its own record (divergent)  1: all lanes of a wave the same record (k_vmem_coalesced: lane-consecutive)
template <int PATTERN, int ADDR>
__global__ __launch_bounds__(256) void k_vmem(const float4 *__restrict__ table, const unsigned *__restrict__ table2, unsigned mask,
                                             int loops, float *__restrict__ sink) {
    const unsigned lane = threadIdx.x & 63u;
    unsigned state = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    if (ADDR == 1) state = blockIdx.x * 977u + (threadIdx.x >> 6) * 131u;
    float acc = 0.0f;
    unsigned acc_u = 0u;
    for (int i = 0; i < loops; i++) {
        state = state * 1664525u + 1013904223u;
        unsigned rec = (state >> 8) & mask;
        const float4 *p = table + (size_t)rec * 4u;
        if (PATTERN == P_DWORD) acc += reinterpret_cast<const float *>(p)[0];
        if (PATTERN == P_X2) { float2 v = reinterpret_cast<const float2 *>(p)[0]; acc += v.x + v.y; }
        if (PATTERN >= P_X4 && PATTERN <= P_4X4) {
            float4 v = p[0];
            acc += v.x + v.w;
            if (PATTERN >= P_2X4) { float4 w = p[1]; acc += w.y + w.w; }
            if (PATTERN >= P_3X4) { float4 w = p[2]; acc += w.z + w.w; }
            if (PATTERN >= P_4X4) { float4 w = p[3]; acc += w.x + w.w; }
        }
        if (PATTERN == P_X4_PLUS_DWORD || PATTERN == P_2X4_PLUS_DWORD) {
            float4 v = p[0];
            acc += v.x + v.w;
            if (PATTERN == P_2X4_PLUS_DWORD) { float4 w = p[1]; acc += w.y + w.w; }
            acc_u += table2[(size_t)rec * 8u + (lane & 7u)];
        }
    }
    if (acc == 1.2345e-30f || acc_u == 0xDEADBEEFu) sink[0] = acc;
}
// (ADDR == 2 wants a wave-uniform base: give every lane of a wave the same LCG state)
template <int PATTERN>
__global__ __launch_bounds__(256) void k_vmem_coalesced(const float4 *__restrict__ table, const unsigned *__restrict__ table2,
                                                       unsigned mask, int loops, float *__restrict__ sink) {
    unsigned lane = threadIdx.x & 63u;
    unsigned state = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (int i = 0; i < loops; i++) {
        state = state * 1664525u + 1013904223u;
        // lane-consecutive 16-byte pieces: a wave reads 1 KiB contiguous per dwordx4
        const float4 *p = table + (size_t)((((state >> 8) & mask) & ~15u) * 4u) + lane;
        float4 v = p[0];
        acc += v.x + v.w;
        if (PATTERN >= P_2X4) { float4 w = p[64]; acc += w.y + w.w; }
        if (PATTERN >= P_4X4) { float4 w = p[128]; acc += w.z; float4 x = p[192]; acc += x.w; }
    }
    if (acc == 1.2345e-30f) sink[0] = acc;
}

typedef void (*Kern)(const float4 *, const unsigned *, unsigned, int, float *);
struct Case { const char *name; Kern fn; int insts; };

int main() {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clock_ghz = prop.clockRate * 1e-6;   // nominal shader clock (the VALU microbenchmark measured 2.38 of 2.40 GHz under load)
    const size_t max_recs = (size_t)1 << 20;          // 64 MiB of 64-byte records
    float4 *d_table;
    unsigned *d_table2;
    float *d_sink;
    CHECK(hipMalloc(&d_table, (max_recs * 4 + 256) * sizeof(float4)));
    CHECK(hipMalloc(&d_table2, max_recs * 8 * sizeof(unsigned)));
    CHECK(hipMalloc(&d_sink, 64));
    CHECK(hipMemset(d_table, 0, (max_recs * 4 + 256) * sizeof(float4)));
    CHECK(hipMemset(d_table2, 0, max_recs * 8 * sizeof(unsigned)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));

    const Case cases[] = {
        {"divergent: dword", k_vmem<P_DWORD, 0>, 1},
        {"divergent: dwordx2", k_vmem<P_X2, 0>, 1},
        {"divergent: dwordx4", k_vmem<P_X4, 0>, 1},
        {"divergent: 2 x dwordx4 of one record", k_vmem<P_2X4, 0>, 2},
        {"divergent: 3 x dwordx4 of one record", k_vmem<P_3X4, 0>, 3},
        {"divergent: 4 x dwordx4 of one record", k_vmem<P_4X4, 0>, 4},
        {"divergent: dwordx4 + dword of a second table", k_vmem<P_X4_PLUS_DWORD, 0>, 2},
        {"divergent: 2 x dwordx4 + dword of a second table", k_vmem<P_2X4_PLUS_DWORD, 0>, 3},
        {"uniform address: dwordx4", k_vmem<P_X4, 1>, 1},
        {"uniform address: 4 x dwordx4", k_vmem<P_4X4, 1>, 4},
        {"coalesced: dwordx4", k_vmem_coalesced<P_X4>, 1},
        {"coalesced: 4 x dwordx4", k_vmem_coalesced<P_4X4>, 4},
    };
    struct Size { const char *name; unsigned mask; } sizes[] = {{"256 KiB", (1u << 12) - 1u}, {"4 MiB", (1u << 16) - 1u}, {"64 MiB", (1u << 20) - 1u}};
    const int waves_per_simd[] = {1, 2, 4, 6};
    const int loops = 2000;
    printf("{\"device\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"clock_ghz_nominal\": %.3f, \"loops\": %d,\n \"results\": [\n", prop.name,
           prop.gcnArchName, cus, clock_ghz, loops);
    bool first = true;
    for (const Size &sz : sizes)
        for (const Case &cs : cases)
            for (int W : waves_per_simd) {
                if (sz.mask != (1u << 16) - 1u && W != 6) continue;   // the other table sizes at full occupancy only
                const int blocks = cus * W;
                float ms = 0, best = 1e30f;
                for (int rep = 0; rep < 3; rep++) {
                    CHECK(hipEventRecord(e0, 0));
                    hipLaunchKernelGGL(cs.fn, dim3(blocks), dim3(256), 0, 0, d_table, d_table2, sz.mask, loops, d_sink);
                    CHECK(hipEventRecord(e1, 0));
                    CHECK(hipEventSynchronize(e1));
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep && ms < best) best = ms;
                }
                CHECK(hipGetLastError());
                const double insts_per_cu = (double)W * 4.0 * loops * cs.insts;   // wave-instructions a CU issued
                const double cycles = best * 1e-3 * clock_ghz * 1e9;
                printf("%s  {\"table\": \"%s\", \"case\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"cyc_per_inst_cu\": %.2f, "
                       "\"cyc_per_iteration_cu\": %.2f}",
                       first ? "" : ",\n", sz.name, cs.name, W, best, cycles / insts_per_cu, cycles / (W * 4.0 * loops));
                first = false;
            }
    printf("\n ]}\n");
    return 0;
}
