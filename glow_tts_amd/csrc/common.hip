// Library identification for libglowtts_hip.so (see include/glowtts_hip.h).
#include <hip/hip_runtime.h>
#include <string.h>
#include "../../include/glowtts_hip.h"

extern "C" int glowtts_abi_version(void) { return 1; }

extern "C" int glowtts_device_arch(char* buf, int buflen)
{
    if (!buf || buflen < 2) return GLOWTTS_E_ARG;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { buf[0] = 0; return GLOWTTS_E_LAUNCH; }
    strncpy(buf, prop.gcnArchName, buflen - 1);
    buf[buflen - 1] = 0;
    return GLOWTTS_OK;
}
