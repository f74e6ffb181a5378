// psd_edge_kernels.hip -- edge term of ContentDetector (placeholder until the Canny pipeline lands).
#include "psd_internal.h"

extern "C" void psd_set_error(const char* fmt, ...);

namespace psd {

int edges_score(psd_engine*, const uint8_t*, int, int, int, size_t, size_t, const uint8_t*, int,
                psd_frame_scores*, hipStream_t)
{
    psd_set_error("PSD_SCORE_EDGES is not implemented yet");
    return PSD_ERR_UNSUPPORTED;
}
int edges_map(psd_engine*, const uint8_t*, int, int, size_t, int, uint8_t*)
{
    psd_set_error("PSD_SCORE_EDGES is not implemented yet");
    return PSD_ERR_UNSUPPORTED;
}
void edges_release(psd_engine*) {}
int resize_linear(const uint8_t*, int, int, int, size_t, uint8_t*, int, int, size_t, hipStream_t)
{
    psd_set_error("psd_resize_linear_device is not implemented yet");
    return PSD_ERR_UNSUPPORTED;
}

}  // namespace psd
