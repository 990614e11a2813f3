// rt_wavefront_launch.inl -- EXPERIMENTS BUILD ONLY: launchers of the wavefront kernels (see rt_wavefront.inl).
// the two kernels of a wavefront sequence (persistent grids; a launch whose list is empty ends at once)
size_t wf_walk_lds_bytes(const RenderArgs& a) {
    return ((size_t)(a.stack_entries ? a.stack_entries : 1u) * (a.stack_wide ? 128u : 64u) + (size_t)a.tlas_entries * 64u) * sizeof(uint32_t) * WAVES_PER_BLOCK;
}
hipError_t launch_wf_shade(const RenderArgs& a, uint32_t blocks, hipStream_t stream) {
    if (blocks == 0) blocks = 1;
    if (a.count_tests) hipLaunchKernelGGL(rt_wf_shade_kernel<true>, dim3(blocks), dim3(BLOCK_THREADS), 0, stream, a);
    else hipLaunchKernelGGL(rt_wf_shade_kernel<false>, dim3(blocks), dim3(BLOCK_THREADS), 0, stream, a);
    return hipGetLastError();
}
hipError_t launch_wf_walk(const RenderArgs& a, uint32_t blocks, hipStream_t stream) {
    if (blocks == 0) blocks = 1;
    const size_t lds = wf_walk_lds_bytes(a);
    if (a.count_tests) launch_k(rt_wf_walk_kernel<true>, blocks, lds, stream, a);
    else launch_k(rt_wf_walk_kernel<false>, blocks, lds, stream, a);
    return hipGetLastError();
}

