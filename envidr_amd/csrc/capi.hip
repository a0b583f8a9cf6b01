// Library-level C-ABI pieces: error text and ABI version.
#include "common.hip.h"

#include <stdarg.h>

namespace envidr {
static thread_local char g_error[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}
}  // namespace envidr

extern "C" {
const char* envidr_last_error(void) { return envidr::g_error; }
int envidr_abi_version(void) { return 10; }   // 10: envidr_release_scratch, fused-pair split-precision kernel (envidr_pack_env_split2, desc.env_split_form); 9: envidr_linear_rows; 8: env-sphere mode (envidr_shell_samples, envidr_composite_shell); 7: layers of <= 16 outputs packed for 16-row MFMA blocks (k_order 2); 6: envidr_geometry_probe, half-precision tables; 5: sdf_geo_blob (16-column geometry kernel); 4: split-precision shading mode
}
