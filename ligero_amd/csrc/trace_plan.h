// host side of the evaluation trace's launch schedule (shared by the context and the kernels' translation units)
#pragma once
#include <cstdint>
#include <vector>

namespace lg {

// how the levels of a program are launched: runs of at least two consecutive narrow levels as one fused launch, the rest one by one
constexpr uint64_t kTraceNarrow = 1024;
struct TraceLaunch { uint32_t level0, level1; bool fused; };
inline std::vector<TraceLaunch> trace_launch_plan(const std::vector<uint64_t>& level_off) {
    std::vector<TraceLaunch> plan;
    const uint32_t n = level_off.empty() ? 0 : (uint32_t)(level_off.size() - 1);
    for (uint32_t l = 0; l < n;) {
        uint32_t e = l;
        while (e < n && level_off[e + 1] - level_off[e] <= kTraceNarrow) e++;
        if (e - l >= 2) { plan.push_back({l, e, true}); l = e; continue; }
        if (level_off[l + 1] > level_off[l]) plan.push_back({l, l + 1, false});
        l++;
    }
    return plan;
}

}  // namespace lg
