// Coordinate conventions of torch.nn.functional.affine_grid / grid_sample as used by the
// reference's stn() (modules.py:265-269); SURVEY.md Appendix A.3.
#pragma once

// Base (normalised) coordinate of output index j of n, in [-1,1].
__host__ __device__ __forceinline__ float stn_base(int j, int n, int align_corners) {
    if (align_corners) return n > 1 ? 2.f * (float)j / (float)(n - 1) - 1.f : 0.f;
    return (2.f * (float)j + 1.f) / (float)n - 1.f;
}

// Source pixel coordinate for output index j: g = scale*base + shift, unnormalised to a source of
// `nsrc` pixels.  With `border`, the coordinate is clipped to [0, nsrc-1] and `mult` (d coord / d g)
// is zeroed outside the clip range exactly like torch's clip_coordinates_set_grad.
// Returns the base coordinate (needed for d/d scale).
// Same as stn_src_coord below with the base coordinate supplied by the caller (a table lookup instead of a division per element).
__host__ __device__ __forceinline__ void stn_src_coord_b(float scale, float shift, float base, int nsrc, int align_corners, bool border,
                                                         float& coord, float& mult) {
    // explicit fused multiply-adds: left to the compiler, the contraction of a*b + c depends on the surrounding code, and two kernels
    // (or a table and a direct evaluation) that must agree on a coordinate would differ in its last bit
    const float g = fmaf(scale, base, shift);
    float c;
    if (align_corners) { c = (g + 1.f) * 0.5f * (float)(nsrc - 1); mult = 0.5f * (float)(nsrc - 1); }
    else { c = fmaf(g + 1.f, (float)nsrc, -1.f) * 0.5f; mult = 0.5f * (float)nsrc; }
    if (border) {
        const float hi = (float)(nsrc - 1);
        if (c <= 0.f) { c = 0.f; mult = 0.f; }
        else if (c >= hi) { c = hi; mult = 0.f; }
    }
    coord = c;
}
__host__ __device__ __forceinline__ float stn_src_coord(float scale, float shift, int j, int nout, int nsrc,
                                                        int align_corners, bool border, float& coord, float& mult) {
    const float base = stn_base(j, nout, align_corners);
    stn_src_coord_b(scale, shift, base, nsrc, align_corners, border, coord, mult);
    return base;
}
