// rt_texture.h -- the canonical definition of
//   textureSampleLevel(textures[i], samplers[0], uv, 0.0)      (wgsl:455,543)
// for the reference's texture setup: Rgba8UnormSrgb, one mip, bilinear
// min/mag filter, repeat addressing (src/rendering/ray_tracer.rs:197-205,253).
// Vulkan leaves the filter's weight precision to the driver (parity unpinned,
// SURVEY.md 8c); this build defines it in full f32:
//   - texel centres at (i + 0.5) / size; sample position p = uv * size - 0.5
//   - i0 = floor(p), f = p - i0, neighbours wrapped with repeat addressing
//   - sRGB -> linear through the exact 256-entry table below (alpha linear)
//   - result = mix(mix(t00, t10, fx), mix(t01, t11, fx), fy), mix(a,b,t) =
//     a*(1-t) + b*t, unfused
// Shared by the HIP kernel and the CPU oracle, like rt_transc.h.
#ifndef RT_TEXTURE_H
#define RT_TEXTURE_H

#include "rt_transc.h"

namespace rtm {

struct TexView {
    const uint8_t* rgba8;  // tightly packed rows
    uint32_t width, height;
};

// Linear value of an sRGB byte; table built on the host by rt_build_srgb_lut
// (double precision formula, rounded once to f32) and passed in.
RT_HD int wrap_repeat(int i, int n) {
    int m = i % n;
    return m < 0 ? m + n : m;
}

RT_HD float floor_(float x) { return __builtin_floorf(x); }

// WP: pointer to 32-bit texels (one RGBA8 texel per word, R in the low byte), LP: pointer to the
// 256-entry sRGB table.  Templates so that the kernel can pass global-address-space pointers
// (global_load instead of flat_load); the oracle passes plain pointers.
template <class WP, class LP>
RT_HD void sample_bilinear_words(WP texels, uint32_t width, uint32_t height, LP srgb_lut, float u, float v, float out[4]) {
    float px = u * (float)width - 0.5f;
    float py = v * (float)height - 0.5f;
    float fx0 = floor_(px), fy0 = floor_(py);
    float fx = px - fx0, fy = py - fy0;
    // NaN / huge coordinates: clamp the integer part so the conversion is defined
    if (!(fx0 > -1.0e9f)) fx0 = -1.0e9f;
    if (!(fx0 < 1.0e9f)) fx0 = 1.0e9f;
    if (!(fy0 > -1.0e9f)) fy0 = -1.0e9f;
    if (!(fy0 < 1.0e9f)) fy0 = 1.0e9f;
    int ix = (int)fx0, iy = (int)fy0;
    int x0 = wrap_repeat(ix, (int)width), x1 = wrap_repeat(ix + 1, (int)width);
    int y0 = wrap_repeat(iy, (int)height), y1 = wrap_repeat(iy + 1, (int)height);
    const uint32_t t00 = texels[(size_t)y0 * width + x0], t10 = texels[(size_t)y0 * width + x1];
    const uint32_t t01 = texels[(size_t)y1 * width + x0], t11 = texels[(size_t)y1 * width + x1];
    for (int c = 0; c < 4; ++c) {
        const uint32_t b00 = (t00 >> (8 * c)) & 255u, b10 = (t10 >> (8 * c)) & 255u;
        const uint32_t b01 = (t01 >> (8 * c)) & 255u, b11 = (t11 >> (8 * c)) & 255u;
        float a, b, cc, d;
        if (c < 3) {
            a = srgb_lut[b00];
            b = srgb_lut[b10];
            cc = srgb_lut[b01];
            d = srgb_lut[b11];
        } else {
            a = (float)b00 / 255.0f;
            b = (float)b10 / 255.0f;
            cc = (float)b01 / 255.0f;
            d = (float)b11 / 255.0f;
        }
        float top = a * (1.0f - fx) + b * fx;
        float bot = cc * (1.0f - fx) + d * fx;
        out[c] = top * (1.0f - fy) + bot * fy;
    }
}

// (texture storage is 4-byte aligned on both sides: hipMalloc / numpy arrays of whole texels)
RT_HD void sample_bilinear(const TexView& t, const float* srgb_lut, float u, float v, float out[4]) {
    if (t.width == 0 || t.height == 0 || t.rgba8 == nullptr) {
        out[0] = out[1] = out[2] = out[3] = 0.0f;
        return;
    }
    sample_bilinear_words(reinterpret_cast<const uint32_t*>(t.rgba8), t.width, t.height, srgb_lut, u, v, out);
}

}  // namespace rtm

#endif
