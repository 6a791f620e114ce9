// hiprtc front end for the specialised pass kernels (rtc.cpp)
#pragma once
#include <string>
#include <vector>

namespace sanm_hip {
//! compile `source` (which may include "program.h" / "tet_ops.h") for gfx950; false + log on failure
bool rtc_compile(const char* source, std::vector<char>& code, std::string& log);
}  // namespace sanm_hip
