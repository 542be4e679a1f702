// wavefront.h — placeholder until the wavefront integrators land (replaced in the next commit)
#pragma once
#include "context.h"
namespace rtx {
struct WavefrontBuffers {};
inline void wavefront_release(rtx_ctx*) {}
inline int wavefront_resize(rtx_ctx*) { return RTX_OK; }
inline int wavefront_render_pass(rtx_ctx*) { return RTX_ERR_STATE; }
inline int wavefront_tile_samples(rtx_ctx*) { return RTX_ERR_STATE; }
}
