// placeholder: filled in by the field milestone
#include "rfx_common.h"
extern "C" {
int rfx_grid_encode_forward(const rfx_grid_desc*, const float*, const float*, int64_t, float*, rfx_stream) { return RFX_ERR_UNSUPPORTED; }
int rfx_grid_encode_backward(const rfx_grid_desc*, const float*, const float*, int64_t, const float*, float*, float*, rfx_stream) { return RFX_ERR_UNSUPPORTED; }
int rfx_oneblob_forward(const float*, int64_t, int, int, float*, rfx_stream) { return RFX_ERR_UNSUPPORTED; }
int rfx_field_forward(const rfx_field_desc*, const float*, int64_t, float*, rfx_stream) { return RFX_ERR_UNSUPPORTED; }
int rfx_field_backward(const rfx_field_desc*, const float*, int64_t, const float*, float*, float*, float*, float*, float*, float*, rfx_stream) { return RFX_ERR_UNSUPPORTED; }
int rfx_field_query_sdf(const rfx_field_desc*, const float*, int64_t, float*, rfx_stream) { return RFX_ERR_UNSUPPORTED; }
int rfx_field_query_color(const rfx_field_desc*, const float*, int64_t, float*, rfx_stream) { return RFX_ERR_UNSUPPORTED; }
}
