// Library identification (include/dcn_hip.h: dcn_version).
#include "dcn_common.h"

extern "C" const char* dcn_version(void) {
#if defined(DCN_HOSTEMU_BUILD)
    return "dcn_hip 0.1 hostemu (TEST BUILD: kernels compiled for the host, tests/hostemu)";
#else
    return "dcn_hip 0.1 gfx950";
#endif
}
