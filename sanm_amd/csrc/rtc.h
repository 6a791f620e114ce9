// Run-time compilation of the pass kernels specialised for a graph (hiprtc bound with dlopen; rtc.cpp).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace sanm_hip {
struct RtcStats {
    int64_t compiled = 0, memory_hits = 0, disk_hits = 0, embedded_hits = 0;
};
//! code object for gfx950 of `source` (which may include "program.h" / "tet_ops.h"); with use_cache the process-wide
//! and the on-disk cache are consulted first and filled afterwards
bool rtc_compile(const char* source, std::vector<char>& code, std::string& log, bool use_cache = true);
RtcStats rtc_stats();
std::string rtc_source_key(const char* source);
void rtc_drop_memory_cache();
}  // namespace sanm_hip
