// Where do the workgroups of a CU-masked stream run?  (tools/gpu_runs/r6_*.sh compile this on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/cu_mask_probe.hip -o /tmp/cu_mask_probe)
// For a list of mask bit ranges [first, first + n): launch 4 096 one-wave workgroups that spin ~20 us each on a stream created with
// hipExtStreamCreateWithCUMask and record (XCC_ID, HW_ID) per workgroup; print the distinct CUs per XCD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <set>
#include <map>

__global__ void probe(uint32_t* out) {
    const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));     // HW_REG_HW_ID
    const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));   // HW_REG_XCC_ID
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) {}                              // 100 MHz counter: 20 us
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int total = prop.multiProcessorCount;
    printf("device: %s, %d CUs\n", prop.name, total);
    const int n_wg = 4096;
    uint32_t* d;
    CK(hipMalloc(&d, n_wg * 8));
    uint32_t* h = (uint32_t*)malloc(n_wg * 8);
    int ranges[][2] = {{0, 256}, {0, 128}, {128, 128}, {0, 32}, {224, 32}, {128, 96}, {144, 80}, {128, 64}, {160, 64}, {0, 8}, {8, 8}, {0, 64}, {64, 64}};
    for (auto& r : ranges) {
        if (r[0] + r[1] > total) continue;
        uint32_t mask[16];
        memset(mask, 0, sizeof(mask));
        for (int i = r[0]; i < r[0] + r[1]; ++i) mask[i >> 5] |= 1u << (i & 31);
        hipStream_t st;
        CK(hipExtStreamCreateWithCUMask(&st, (total + 31) / 32, mask));
        CK(hipMemsetAsync(d, 0xff, n_wg * 8, st));
        hipLaunchKernelGGL(probe, dim3(n_wg), dim3(64), 0, st, d);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(h, d, n_wg * 8, hipMemcpyDeviceToHost));
        std::map<int, std::set<int>> per_xcc;
        std::map<int, int> wg_per_xcc;
        for (int b = 0; b < n_wg; ++b) {
            const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
            const int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
            wg_per_xcc[xcc]++;
        }
        int sum = 0;
        printf("mask bits [%3d, %3d): CUs seen per XCD:", r[0], r[0] + r[1]);
        for (auto& kv : per_xcc) { printf(" x%d=%zu", kv.first, kv.second.size()); sum += (int)kv.second.size(); }
        printf("  (total %d)  workgroups per XCD:", sum);
        for (auto& kv : wg_per_xcc) printf(" %d", kv.second);
        printf("\n");
        if (r[1] <= 32) {
            printf("    XCD 0 CUs (se.sh.cu):");
            for (int v : per_xcc[0]) printf(" %d.%d.%d", v >> 8, (v >> 4) & 1, v & 15);
            printf("\n");
        }
        CK(hipStreamDestroy(st));
    }
    return 0;
}
