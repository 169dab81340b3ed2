// placement_probe.hip — where the dispatcher puts the workgroups and waves of a launch shaped like the chained Jacobi launch
// (512-thread workgroups, two per CU), read from inside the kernel: HW_REG_XCC_ID, HW_REG_HW_ID (SE / CU / SIMD / wave slot) and the
// 100 MHz wall clock at start and end.  Answers, for speed decisions only (HIP promises none of it, MI355X guide "Workgroup dispatch"):
//   * is workgroup b on XCD b % 8;  * in which order do the workgroups of one XCD start (is it blockIdx order);
//   * which SIMDs do the eight waves of a workgroup land on, and do two co-resident workgroups complement each other.
// Build: hipcc --offload-arch=gfx950 -O3 -o placement_probe placement_probe.hip ; run: ./placement_probe [workgroups] [spin us]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Rec {
    unsigned hw_id, xcc;
    unsigned long long t0, t1;
};

__global__ void __launch_bounds__(512, 4) k_probe(Rec* out, unsigned spin_ticks)
{
    __shared__ float pad[18 * 1024];   // 72 KiB: two workgroups per CU, like the Jacobi tile kernel
    const int wave = threadIdx.y;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID, 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);   // HW_REG_XCC_ID, bits 3:0
    pad[threadIdx.y * 64 + threadIdx.x] = (float)hw;
    __syncthreads();
    float acc = pad[(threadIdx.x * 7 + wave) & 511];
    while (__builtin_amdgcn_s_memrealtime() - t0 < spin_ticks) acc = acc * 1.0001f + 1.0f;   // stay resident: co-residency shows in the times
    if (threadIdx.x == 0) {
        Rec r;
        r.hw_id = hw;
        r.xcc = xcc;
        r.t0 = t0;
        r.t1 = __builtin_amdgcn_s_memrealtime();
        out[blockIdx.x * 8 + wave] = r;
    }
    if (acc == 12345.678f) out[0].hw_id = 0;
}

int main(int argc, char** argv)
{
    const int nwg = argc > 1 ? atoi(argv[1]) : 3000;
    const unsigned spin_us = argc > 2 ? atoi(argv[2]) : 15;
    Rec* d;
    CK(hipMalloc(&d, sizeof(Rec) * nwg * 8));
    std::vector<Rec> h(nwg * 8);
    for (int rep = 0; rep < 2; rep++) {
        CK(hipMemset(d, 0, sizeof(Rec) * nwg * 8));
        k_probe<<<dim3(nwg), dim3(64, 8)>>>(d, spin_us * 100);
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h.data(), d, sizeof(Rec) * nwg * 8, hipMemcpyDeviceToHost));
    auto simd = [](unsigned hw) { return (hw >> 4) & 3; };
    auto slot = [](unsigned hw) { return hw & 15; };
    auto cu = [](unsigned hw) { return (hw >> 8) & 15; };
    auto se = [](unsigned hw) { return (hw >> 13) & 7; };
    // 1. XCD of workgroup b
    int on_mod8 = 0;
    std::map<int, int> xcc_hist;
    for (int b = 0; b < nwg; b++) {
        on_mod8 += (int)(h[b * 8].xcc & 7) == b % 8;
        xcc_hist[h[b * 8].xcc & 15]++;
    }
    printf("workgroups %d, spin %u us\n", nwg, spin_us);
    printf("1. workgroup b on XCD b %% 8: %d of %d;  XCC_ID histogram:", on_mod8, nwg);
    for (auto& kv : xcc_hist) printf(" %d:%d", kv.first, kv.second);
    printf("\n");
    // 2. start order per XCD: sort the XCD's workgroups by start time, compare with blockIdx order
    unsigned long long tmin = ~0ull;
    for (auto& r : h) tmin = std::min(tmin, r.t0);
    for (int x = 0; x < 8; x++) {
        std::vector<std::pair<unsigned long long, int>> v;
        for (int b = 0; b < nwg; b++)
            if ((int)(h[b * 8].xcc & 7) == x) v.push_back({ h[b * 8].t0, b });
        std::stable_sort(v.begin(), v.end());
        int inversions = 0, maxdisp = 0;
        std::vector<int> ids;
        for (auto& p : v) ids.push_back(p.second);
        std::vector<int> sorted = ids;
        std::sort(sorted.begin(), sorted.end());
        for (size_t i = 0; i < ids.size(); i++) {
            const int want = (int)(std::lower_bound(sorted.begin(), sorted.end(), ids[i]) - sorted.begin());
            maxdisp = std::max(maxdisp, abs(want - (int)i));
            if (i + 1 < ids.size() && ids[i + 1] < ids[i] && v[i + 1].first > v[i].first) inversions++;
        }
        printf("2. XCD %d: %zu workgroups; started out of blockIdx order (strictly later clock, lower id): %d; largest displacement from id order: %d places; first start +%.2f us, last start +%.2f us\n",
               x, ids.size(), inversions, maxdisp, v.empty() ? 0.0 : (v.front().first - tmin) / 100.0, v.empty() ? 0.0 : (v.back().first - tmin) / 100.0);
    }
    // 3. SIMDs of a workgroup's eight waves
    std::map<std::string, int> pat;
    for (int b = 0; b < nwg; b++) {
        std::string s;
        for (int w = 0; w < 8; w++) s += (char)('0' + simd(h[b * 8 + w].hw_id));
        pat[s]++;
    }
    printf("3. SIMD of waves 0..7 of a workgroup (pattern: count):");
    for (auto& kv : pat) printf("  %s:%d", kv.first.c_str(), kv.second);
    printf("\n");
    std::map<std::string, int> slots;
    for (int b = 0; b < nwg; b++) {
        std::string s;
        for (int w = 0; w < 8; w++) s += (char)('0' + slot(h[b * 8 + w].hw_id) % 10);
        slots[s]++;
    }
    printf("   wave-slot ids (HW_ID 3:0) of waves 0..7, the eight most frequent patterns:");
    std::vector<std::pair<int, std::string>> sv;
    for (auto& kv : slots) sv.push_back({ kv.second, kv.first });
    std::sort(sv.rbegin(), sv.rend());
    for (size_t i = 0; i < sv.size() && i < 8; i++) printf("  %s:%d", sv[i].second.c_str(), sv[i].first);
    printf("\n");
    // 4. co-residency: workgroups on the same (xcc, se, cu) whose lifetimes overlap
    std::map<unsigned, std::vector<int>> by_cu;
    for (int b = 0; b < nwg; b++) by_cu[(h[b * 8].xcc & 7) << 8 | se(h[b * 8].hw_id) << 4 | cu(h[b * 8].hw_id)].push_back(b);
    int pairs = 0, same_first_simd = 0, max_resident = 0;
    std::map<std::string, int> pairpat;
    for (auto& kv : by_cu) {
        auto& v = kv.second;
        for (size_t i = 0; i < v.size(); i++) {
            int resident = 1;
            for (size_t j = 0; j < v.size(); j++) {
                if (i == j) continue;
                const Rec &a = h[v[i] * 8], &c = h[v[j] * 8];
                if (c.t0 <= a.t0 && a.t0 < c.t1) {   // c was resident when a started
                    resident++;
                    pairs++;
                    same_first_simd += simd(a.hw_id) == simd(c.hw_id);
                    std::string s;
                    s += (char)('0' + simd(a.hw_id));
                    s += (char)('0' + simd(c.hw_id));
                    pairpat[s]++;
                }
            }
            max_resident = std::max(max_resident, resident);
        }
    }
    printf("4. CUs seen: %zu; largest number of workgroups resident together on a CU: %d; (new, resident) pairs: %d, first wave on the same SIMD in %d;  (SIMD of wave 0: new, resident):",
           by_cu.size(), max_resident, pairs, same_first_simd);
    for (auto& kv : pairpat) printf(" %s:%d", kv.first.c_str(), kv.second);
    printf("\n");
    printf("   raw HW_ID of workgroup 0's waves:");
    for (int w = 0; w < 8; w++) printf(" %08x", h[w].hw_id);
    printf("\n");
    return 0;
}
