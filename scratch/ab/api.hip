// ABI version and error strings of libddnerf_hip.so.
#include "common.h"

DDN_EXPORT int ddnerf_abi_version(void) { return 1; }

DDN_EXPORT const char *ddnerf_error_string(int code) {
    switch (code) {
        case DDNERF_OK: return "ok";
        case DDNERF_E_ARG: return "ddnerf: null pointer or non-positive size";
        case DDNERF_E_RANGE: return "ddnerf: size outside the range the kernel supports";
        case DDNERF_E_ALIGN: return "ddnerf: pointer not 16-byte aligned";
        case DDNERF_E_WORKSPACE: return "ddnerf: workspace too small";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "ddnerf: unknown error";
    }
}
