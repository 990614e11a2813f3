// rt_api_wavefront.inl -- host side of the wavefront sequence (option "wavefront"): built, parity-tested,
// measured slower than the inline kernels; compiled with -DRT_EXPERIMENTS=1 only (included by rt_api.hip).  Moved out of
// render_impl in round 5 as it was (code motion, no change).

// Which launches take the wavefront sequence, and its buffers: one slot per pixel and frame of the launch (whole 8x8 tiles); a
// path makes at most rays_per_pixel x (bounces + 1) traversals, one per round.
static int wavefront_prepare(rt_handle* h, const rt_params* params, const RenderArgs& a, uint32_t n_tiles, uint32_t n_batch,
                             WavefrontPlan& wf, bool& rounds) {
    const uint32_t wf_frame_slots = n_tiles * 64u;
    const uint64_t wf_slots64 = (uint64_t)wf_frame_slots * (n_batch ? n_batch : 1u);
    const uint64_t wf_rounds64 = params->number_of_bounces < 0 ? 0ull
        : (uint64_t)(params->rays_per_pixel > 0 ? params->rays_per_pixel : 0) * ((uint64_t)params->number_of_bounces + 1ull);
    bool wavefront = h->wavefront != 0 && a.many_mesh != 0 && !h->any_deep && params->debug_flag == 0 && params->rays_per_pixel > 0 &&
                     params->width <= 0xffffu && params->height <= 0xffffu && wf_slots64 < (1ull << 31) && wf_rounds64 <= 1024ull;
    const size_t wf_bytes_per_slot = (WF_STATE_PLANES + WF_HIT_PLANES) * sizeof(float4) + 2 * sizeof(uint32_t) + PIXEL_MEMO_DWORDS * sizeof(uint32_t);
    if (wavefront && h->wf_capacity < wf_slots64) {
        size_t free_b = 0, total_b = 0;
        const size_t want = (size_t)wf_slots64 * wf_bytes_per_slot;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || want > free_b / 2)
            return fail(h, RT_ERR_OUT_OF_MEMORY, "wavefront: the path slots of this launch do not fit half of the free device memory");
    }
    if (wavefront) {
        const size_t blocks64 = ((size_t)wf_slots64 + 63) / 64;
        if (h->wf_capacity < wf_slots64) {
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            free_dev(h->wf_state);
            free_dev(h->wf_hit);
            free_dev(h->wf_lists);
            h->wf_capacity = 0;
            hipError_t e = hipMalloc((void**)&h->wf_state, blocks64 * WF_STATE_PLANES * 64 * sizeof(float4));
            if (e == hipSuccess) e = hipMalloc((void**)&h->wf_hit, blocks64 * WF_HIT_PLANES * 64 * sizeof(float4));
            if (e == hipSuccess) e = hipMalloc((void**)&h->wf_lists, 2 * blocks64 * 64 * sizeof(uint32_t));
            if (e != hipSuccess) {
                (void)hipGetLastError();
                free_dev(h->wf_state);
                free_dev(h->wf_hit);
                free_dev(h->wf_lists);
                if (h->wavefront > 0) return fail(h, RT_ERR_OUT_OF_MEMORY, "wavefront: no device memory for the path slots");
                wavefront = false;
            } else {
                h->wf_capacity = blocks64 * 64;
                // (a hit record is only read after the walk kernel has written it; zeroed all the same, so that a slot
                // the sequence mishandled would read as a miss instead of as indices into nowhere)
                HIP_TRY(h, hipMemsetAsync(h->wf_state, 0, blocks64 * WF_STATE_PLANES * 64 * sizeof(float4), h->stream));
                HIP_TRY(h, hipMemsetAsync(h->wf_hit, 0, blocks64 * WF_HIT_PLANES * 64 * sizeof(float4), h->stream));
                HIP_TRY(h, hipMemsetAsync(h->wf_lists, 0, 2 * blocks64 * 64 * sizeof(uint32_t), h->stream));
            }
        }
    }
    if (wavefront) {
        const size_t need_counts = (size_t)wf_rounds64 + 2;
        if (h->wf_counts_capacity < need_counts) {
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            free_dev(h->wf_counts);
            HIP_TRY(h, hipMalloc((void**)&h->wf_counts, need_counts * sizeof(uint32_t)));
            h->wf_counts_capacity = need_counts;
        }
        // the slots' primary-ray memos: 13 dwords each, in blocks of 64 slots
        const size_t need = (((size_t)wf_slots64 + 63) / 64) * 64 * PIXEL_MEMO_DWORDS;
        if (h->pixel_cache_words < need) {
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            free_dev(h->pixel_cache_mem);
            h->pixel_cache_words = 0;
            HIP_TRY(h, hipMalloc((void**)&h->pixel_cache_mem, need * sizeof(uint32_t)));
            h->pixel_cache_words = need;
        }
        rounds = false;
    }
    wf.on = wavefront;
    wf.frame_slots = wf_frame_slots;
    wf.slots = wf_slots64;
    wf.rounds = wf_rounds64;
    return RT_OK;
}

// shade launch 0 takes the pixels; then, round after round, the walk kernel intersects the scene for the listed rays and the
// shade kernel finishes those segments and lists the next ones.  Everything is ordered on the stream; a launch whose list is
// empty ends at once.
static int wavefront_run(rt_handle* h, const RenderArgs& a, const WavefrontPlan& wf) {
    const uint64_t wf_rounds64 = wf.rounds, wf_slots64 = wf.slots;
    const uint32_t wf_frame_slots = wf.frame_slots;
    // shade launch 0 takes the pixels; then, round after round, the walk kernel intersects the scene for the listed
    // rays and the shade kernel finishes those segments and lists the next ones.  Everything is ordered on the
    // stream; a launch whose list is empty ends at once.
    const uint32_t R = (uint32_t)wf_rounds64;
    HIP_TRY(h, hipMemsetAsync(h->wf_counts, 0, ((size_t)R + 2) * sizeof(uint32_t), h->stream));
    RenderArgs w = a;
    w.pixel_cache = h->pixel_cache_opt ? 3u : 0u;   // (0: no memo; every sample traverses)
    w.pixel_cache_mem = h->pixel_cache_mem;
    if (!w.pixel_cache) w.primary = nullptr;
    w.tile_order = nullptr;
    w.tile_cost = nullptr;
    w.top_count = 0;
    w.tlas_lds = 0;
    w.wf_state = h->wf_state;
    w.wf_hit = h->wf_hit;
    w.wf_slots = (uint32_t)wf_slots64;
    w.wf_frame_slots = wf_frame_slots;
    uint32_t* lists[2] = {h->wf_lists, h->wf_lists + h->wf_capacity};
    const uint32_t shade_blocks = h->compute_units * 4u;  // (the shade kernel is compiled for 4 waves per SIMD: 128 VGPRs)
    const size_t wlds = wf_walk_lds_bytes(w);
    uint32_t walk_per_cu = wlds ? (uint32_t)((160u * 1024u) / wlds) : BLOCKS_PER_CU;
    if (walk_per_cu > BLOCKS_PER_CU) walk_per_cu = BLOCKS_PER_CU;
    if (walk_per_cu < 1u) walk_per_cu = 1u;
    const uint32_t walk_blocks = h->compute_units * walk_per_cu;
    w.wf_round0 = 1;
    w.wf_list_in = nullptr;
    w.wf_count_in = nullptr;
    w.wf_list_out = lists[0];
    w.wf_count_out = h->wf_counts;
    HIP_TRY(h, launch_wf_shade(w, shade_blocks, h->stream));
    w.wf_round0 = 0;
    for (uint32_t r = 0; r < R; ++r) {
        w.wf_list_in = lists[r & 1u];
        w.wf_count_in = h->wf_counts + r;
        h->work_slot = (h->work_slot + 1) & 63u;
        if (h->work_slot == 0u) HIP_TRY(h, hipMemsetAsync(h->work_counters, 0, 64 * sizeof(uint32_t), h->stream));
        w.work_counter = h->work_counters + h->work_slot;
        HIP_TRY(h, launch_wf_walk(w, walk_blocks, h->stream));
        w.wf_list_out = lists[(r + 1) & 1u];
        w.wf_count_out = h->wf_counts + r + 1;
        HIP_TRY(h, launch_wf_shade(w, shade_blocks, h->stream));
    }
    h->last_launch[0] = (uint32_t)wlds;
    h->last_launch[1] = walk_blocks;
    h->last_launch[3] |= 16u;
    return RT_OK;
}
