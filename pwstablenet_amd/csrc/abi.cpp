// Version / error plumbing of the C ABI (include/pwstable.h).
#include <vector>

#include "common.h"

namespace pws {
bool g_two_queues = true;
int g_math = PWS_MATH_FP32;
int g_store = PWS_STORE_FP32;
int g_experiment = 0;
bool g_prof_on = false;
thread_local int g_prof_tag = -1;
thread_local bool t_deterministic = false;
namespace {
struct ProfEntry {
    hipEvent_t a, b;
    int kernel_id, tag;
    double flops, bytes;
};
std::vector<ProfEntry> g_prof_log;
std::vector<hipEvent_t> g_prof_pool;
hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) {
        hipEvent_t e = g_prof_pool.back();
        g_prof_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
const char *const kKernelNames[KID_COUNT] = {
    "conv_mfma_kernel<k3s1,16x16>", "conv_mfma_kernel<k3s1,small>", "conv_mfma_kernel<k3s2,16x16>",
    "conv_mfma_kernel<k3s2,small>", "conv_mfma_kernel<k5s1,16x16>", "conv_mfma_kernel<convT4,16x16>",
    "conv_mfma_kernel<convT4,small>", "theta_head_kernel", "field_head_kernel", "grid_sample_fwd_kernel",
    "grid_sample_bwd_kernel", "upsample_grid_sample_fwd_kernel", "upsample_bilinear_ac_kernel", "affine_grid_kernel",
    "adam_kernel", "pack_weight_kernel", "conv_mfma_kernel<dgrad k4s2>", "conv_mfma_kernel<dgrad subpix k3s2>",
    "wgrad_mfma_kernel", "act_bwd_bias_kernel", "field_head_bwd_kernels", "theta_head_bwd_kernels",
    "wino_k3s1_kernel<F(2x2,3x3)>", "conv_bf16_kernel", "wgrad_bf16_kernel",
    "wino_ct4_kernel<F(3x3,2x2)>", "upsample_grid_sample_u8_kernel", "objective_kernels", "conv_ring_kernel", "conv_ringf_kernel",
    "wino_ring_kernel<F(2x2,3x3)>", "wino_ring_kernel<convT4,F(2x2,2x2)>", "conv_skinny_kernel", "wgrad_ring_kernel", "conv_skinny16_kernel", "conv_first_kernel", "wino5_first_kernel"};
}  // namespace

bool prof_begin(int kernel_id, double flops, double bytes, hipStream_t st) {
    // an event recorded while the stream is captured becomes a graph node, not a timestamp (hipEventElapsedTime: invalid resource handle)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return false;
    ProfEntry e{prof_event(), prof_event(), kernel_id, g_prof_tag, flops, bytes};
    (void)hipEventRecord(e.a, st);
    g_prof_log.push_back(e);
    return true;
}
void prof_end(hipStream_t st) {
    if (!g_prof_log.empty()) (void)hipEventRecord(g_prof_log.back().b, st);
}
}  // namespace pws

namespace pws {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace pws

extern "C" int pws_version(void) { return PWS_VERSION; }

extern "C" const char *pws_last_error(void) { return pws::g_err; }

extern "C" int pws_device_info(int *compute_units, int *arch_is_gfx950) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) {
        pws::set_error("pws_device_info: %s", hipGetErrorString(e));
        return PWS_EHIP;
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (arch_is_gfx950) {
        const char *a = prop.gcnArchName;
        *arch_is_gfx950 = (a[0] == 'g' && a[1] == 'f' && a[2] == 'x' && a[3] == '9' && a[4] == '5' && a[5] == '0') ? 1 : 0;
    }
    return PWS_OK;
}

extern "C" int pws_prof_enable(int on) {
    pws::g_prof_on = on != 0;
    return PWS_OK;
}

extern "C" int pws_prof_collect(pws_prof_record *out, int max_records) {
    const int n = (int)pws::g_prof_log.size();
    hipError_t bad = hipSuccess;
    for (int i = 0; i < n; ++i) {
        pws::ProfEntry &e = pws::g_prof_log[i];
        float ms = 0.f;
        hipError_t err = hipEventSynchronize(e.b);
        if (err == hipSuccess) err = hipEventElapsedTime(&ms, e.a, e.b);
        if (err != hipSuccess && bad == hipSuccess) bad = err;
        if (err == hipSuccess && out && i < max_records) out[i] = pws_prof_record{e.kernel_id, e.tag, e.flops, e.bytes, ms};
        pws::g_prof_pool.push_back(e.a);
        pws::g_prof_pool.push_back(e.b);
    }
    pws::g_prof_log.clear();   // (also after an error: the next collection starts clean)
    if (bad != hipSuccess) {
        pws::set_error("pws_prof_collect: %s", hipGetErrorString(bad));
        return PWS_EHIP;
    }
    return n;
}

extern "C" const char *pws_prof_kernel_name(int kernel_id) {
    return (kernel_id >= 0 && kernel_id < pws::KID_COUNT) ? pws::kKernelNames[kernel_id] : "?";
}

extern "C" int pws_set_option(int key, int value) {
    if (key == PWS_OPT_TWO_QUEUES) {
        pws::g_two_queues = value != 0;
        return PWS_OK;
    }
    if (key == PWS_OPT_MATH) {
        if (value != PWS_MATH_FP32 && value != PWS_MATH_BF16) {
            pws::set_error("pws_set_option: PWS_OPT_MATH value %d is not PWS_MATH_FP32 / PWS_MATH_BF16", value);
            return PWS_EINVAL;
        }
        pws::g_math = value;
        return PWS_OK;
    }
    if (key == PWS_OPT_STORE) {
        if (value != PWS_STORE_FP32 && value != PWS_STORE_BF16) {
            pws::set_error("pws_set_option: PWS_OPT_STORE value %d is not PWS_STORE_FP32 / PWS_STORE_BF16", value);
            return PWS_EINVAL;
        }
        pws::g_store = value;
        return PWS_OK;
    }
    if (key == PWS_OPT_EXPERIMENT) {
        pws::g_experiment = value;
        return PWS_OK;
    }
    pws::set_error("pws_set_option: unknown key %d", key);
    return PWS_EINVAL;
}

extern "C" int pws_get_option(int key) {
    if (key == PWS_OPT_TWO_QUEUES) return pws::g_two_queues ? 1 : 0;
    if (key == PWS_OPT_MATH) return pws::g_math;
    if (key == PWS_OPT_STORE) return pws::g_store;
    if (key == PWS_OPT_EXPERIMENT) return pws::g_experiment;
    pws::set_error("pws_get_option: unknown key %d", key);
    return PWS_EINVAL;
}
