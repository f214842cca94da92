// ABI version and error strings of libddnerf_hip.so.
#include "common.h"
#include <cstdio>

DDN_EXPORT int ddnerf_abi_version(void) { return 1; }

// (mlp_bf16_g2.hip: the experiment switches its generated tile body was produced with; "" = the product body)
extern "C" const char *ddnerf_bf16g2_generator_options(void);

DDN_EXPORT const char *ddnerf_build_info(void) {
    static const struct Info {
        char text[512];
        Info() {
#ifdef BF16_STAMP
            const char *kind = "diagnostic build (clock stamps in the bf16 kernels)";
#else
            const char *kind = "product build";
#endif
            snprintf(text, sizeof(text), "libddnerf_hip abi 1, gfx950, %s; bf16 two-group body generator options: \"%s\"", kind,
                     ddnerf_bf16g2_generator_options());
        }
    } info;
    return info.text;
}

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
