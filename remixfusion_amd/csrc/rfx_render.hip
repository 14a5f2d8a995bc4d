// placeholder: filled in by the render milestone
#include "rfx_common.h"
extern "C" {
int rfx_sample_z(const rfx_sampler_desc*, const float*, const float*, int64_t, float*, rfx_stream) { return RFX_ERR_UNSUPPORTED; }
int rfx_ray_points(const float*, const float*, const float*, int64_t, int, const float*, float*, rfx_stream) { return RFX_ERR_UNSUPPORTED; }
int rfx_composite_forward(const float*, const float*, int64_t, int, float, float, float*, float*, float*, rfx_stream) { return RFX_ERR_UNSUPPORTED; }
int rfx_composite_backward(const float*, const float*, int64_t, int, float, float, const float*, const float*, float*, rfx_stream) { return RFX_ERR_UNSUPPORTED; }
int rfx_render_rays(const rfx_field_desc*, const rfx_sampler_desc*, const float*, const float*, const float*, int64_t, const float*, float, float*, float*, rfx_stream) { return RFX_ERR_UNSUPPORTED; }
}
