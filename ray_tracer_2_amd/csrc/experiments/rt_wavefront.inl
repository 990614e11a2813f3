// rt_wavefront.inl -- EXPERIMENTS BUILD ONLY (-DRT_EXPERIMENTS=1; included by rt_kernel.hip inside namespace rtd).
// The wavefront sequence of round 3 (option "wavefront"): path state in memory slots, a shading kernel and a ray-walk
// kernel with per-lane refill alternating.  Built, bit-identical incl. the counters, measured SLOWER than the inline
// many-mesh kernels (11.3 against 10.2 ms per frame on the sponza-sized stand-in, before cross-mesh
// pruning took the inline kernels to 5.6) -- kept for the record and its parity tests, not part of the product library.
// ---------------------------------------------------------------------------
// Wavefront sequence (RenderArgs::wf_*, rt_device.h): the path state of every pixel of the launch lives in a slot in
// global memory; rt_wf_shade_kernel runs everything of path_step but the traversal, rt_wf_walk_kernel runs the
// traversal (intersect_scene's many-mesh form) for the listed rays.
// ---------------------------------------------------------------------------
namespace {

DEV float4* wf_plane(float4* base, uint32_t planes, uint32_t slot) {
    return base + (size_t)(slot >> 6) * (planes * 64u) + (slot & 63u);
}

DEV void wf_store_state(const RenderArgs& a, uint32_t slot, const PixelState& s) {
    float4* q = wf_plane(a.wf_state, WF_STATE_PLANES, slot);
    auto u = [](uint32_t v) { return __uint_as_float(v); };
    q[0 * 64] = make_float4(u(s.x | (s.out_row << 16)), u(s.rng), u((uint32_t)s.j | (s.fresh ? 0x80000000u : 0u)), u((uint32_t)s.seg));
    q[1 * 64] = make_float4(s.ro.x, s.ro.y, s.ro.z, s.rd.x);
    q[2 * 64] = make_float4(s.rd.y, s.rd.z, s.T.x, s.T.y);
    q[3 * 64] = make_float4(s.T.z, s.T.w, s.light.x, s.light.y);
    q[4 * 64] = make_float4(s.light.z, s.light.w, s.total.x, s.total.y);
    q[5 * 64] = make_float4(s.total.z, s.total.w, u(s.meta), 0.0f);
}

DEV void wf_load_state(const RenderArgs& a, uint32_t slot, PixelState& s) {
    const float4* q = wf_plane(a.wf_state, WF_STATE_PLANES, slot);
    const float4 p0 = q[0 * 64], p1 = q[1 * 64], p2 = q[2 * 64], p3 = q[3 * 64], p4 = q[4 * 64], p5 = q[5 * 64];
    s.x = fbits(p0.x) & 0xffffu;
    s.out_row = fbits(p0.x) >> 16;
    s.rng = fbits(p0.y);
    s.j = (int32_t)(fbits(p0.z) & 0x7fffffffu);
    s.fresh = (fbits(p0.z) & 0x80000000u) != 0u;
    s.seg = (int32_t)fbits(p0.w);
    s.ro = f3{p1.x, p1.y, p1.z};
    s.rd = f3{p1.w, p2.x, p2.y};
    s.T = f4{p2.z, p2.w, p3.x, p3.y};
    s.light = f4{p3.z, p3.w, p4.x, p4.y};
    s.total = f4{p4.z, p4.w, p5.x, p5.y};
    s.meta = fbits(p5.z);
}

DEV void wf_store_hit(const RenderArgs& a, uint32_t slot, const Isect& I) {
    float4* q = wf_plane(a.wf_hit, WF_HIT_PLANES, slot);
    const uint32_t code = I.object >= 0 ? (uint32_t)I.object : (0x800000u | (uint32_t)(-I.object - 1));
    const uint32_t word = code | (I.any ? 1u << 24 : 0u) | (I.s_inside ? 1u << 25 : 0u);
    q[0] = make_float4(I.closest, __uint_as_float(word), I.win_u, I.win_v);
    q[64] = make_float4(__uint_as_float(I.win_tri), I.win_point.x, I.win_point.y, I.win_point.z);
}

DEV void wf_load_hit(const RenderArgs& a, uint32_t slot, Isect& I) {
    const float4* q = wf_plane(a.wf_hit, WF_HIT_PLANES, slot);
    const float4 h0 = q[0], h1 = q[64];
    const uint32_t word = fbits(h0.y);
    I.closest = h0.x;
    I.object = (word & 0x800000u) ? -(int)(word & 0x7fffffu) - 1 : (int)(word & 0x7fffffu);
    I.any = (word & (1u << 24)) != 0u;
    I.s_inside = (word & (1u << 25)) != 0u;
    I.s_dst = h0.x;  // (a sphere winner's distance is the closest distance, isect_spheres)
    I.win_u = h0.z;
    I.win_v = h0.w;
    I.win_tri = fbits(h1.x);
    I.win_point = f3{h1.y, h1.z, h1.w};
}

}  // namespace

// Everything of path_step except the traversal, for the slots of wf_list_in (round 0: for every slot, which takes its
// pixel): finish the segment whose closest-hit record the walk kernel left in wf_hit (isect_finish, the memo, path_end),
// then go on -- next sample, memoised primary segments, ends of paths -- until the slot's next segment needs a
// traversal (it is listed in wf_list_out) or its pixel is finished (the texel is stored).
template <bool STATS>
__global__ void __launch_bounds__(BLOCK_THREADS, 4) rt_wf_shade_kernel(const RenderArgs a) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6), n_waves = gridDim.x * WAVES_PER_BLOCK;
    const bool round0 = a.wf_round0 != 0u;
    const uint32_t count = round0 ? a.wf_slots : *a.wf_count_in;
    const bool have_samples = a.params.rays_per_pixel > 0;
    uint32_t n_segments = 0, n_reused = 0;
    for (uint32_t base = wave * 64u; base < count; base += n_waves * 64u) {  // (wave-uniform)
        const uint32_t i = base + lane;
        const bool valid = i < count;
        const uint32_t slot = valid ? (round0 ? i : a.wf_list_in[i]) : 0u;
        uint32_t* memo = a.pixel_cache_mem + (size_t)(slot >> 6) * (PIXEL_MEMO_DWORDS * 64u) + (slot & 63u);
        PixelState s;
        bool active = false, want = false;
        if (valid) {
            if (round0) {
                const uint32_t k = slot / a.wf_frame_slots, q = slot - k * a.wf_frame_slots;  // frame of the batch, pixel
                const PixelCoord px = pixel_of(a, q >> 6, q & 63u);
                if (px.valid) {
                    const CameraConsts cam = camera_consts(a);
                    pixel_begin<false>(a, cam, s, nullptr, px.x, px.y, px.out_row, k);
                    pixel_cache_begin<true>(a, a, cam, s, memo);
                    s.meta = k << 19;
                    if (have_samples) active = true;
                    else pixel_finish<false>(a, s, nullptr);  // 0 / 0 = NaN, as the shader would store
                }
            } else {
                wf_load_state(a, slot, s);
                active = true;
            }
        }
        if (active) {
            bool done = false;
            if (!round0) {
                Isect I;
                wf_load_hit(a, slot, I);
                const Hit hit = isect_finish<false>(a, I, s.ro, s.rd);
                memo_hit_store<STATS, true>(a, s, memo, hit);
                done = path_end<false, false>(a, s, nullptr, STEP_TRAVERSE, hit, n_segments);
            }
            while (!done) {
                uint32_t starve = 0;
                const uint32_t mode = path_begin<STATS, false, true>(a, s, memo, starve);
                if (mode == STEP_TRAVERSE) {
                    want = true;
                    break;
                }
                Hit hit;
                hit.hit = false;
                hit.suspended = false;
                if (mode == STEP_REUSE) {
                    memo_hit_load<true>(a, s, memo, hit);
                    n_reused += 1;
                }
                done = path_end<false, false>(a, s, nullptr, mode, hit, n_segments);
            }
            if (done) pixel_finish<false>(a, s, nullptr);
            if (want) wf_store_state(a, slot, s);
        }
        // list the slots that need a traversal: consecutive entries, one counter update per wave
        const unsigned long long m = __ballot(want);
        if (m != 0ull) {
            uint32_t b0 = 0;
            if (lane == 0u) b0 = atomicAdd(a.wf_count_out, (uint32_t)__popcll(m));
            b0 = __builtin_amdgcn_readfirstlane(b0);
            if (want) a.wf_list_out[b0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = slot;
        }
    }
    if (a.counters && __ballot(n_segments != 0u || n_reused != 0u) != 0ull) {
        atomicAdd(&a.counters->segments, (unsigned long long)n_segments);
        atomicAdd(&a.counters->reused, (unsigned long long)n_reused);
    }
}

// intersect_scene's many-mesh form (wgsl:353-396: spheres, then the mesh loop as items -- single meshes with root-box
// culling, top-level trees over mesh root boxes) for the rays of wf_list_in, one ray per lane, as a per-lane state
// machine: a lane is idle, or needs a BOX step (one tree node or one mesh node: two slab tests, push / pop), or a LEAF
// step (the triangles of the leaf in hand, then a pop), or an ADVANCE step (offer the hit of the mesh just left,
// start the next item, write the result).  Each pass of the loop runs ONE kind of step, the one most lanes are
// waiting for; idle lanes take the next rays of the list.  Per ray the sequence of tests -- per mesh traverse_mesh's,
// per tree the iterator's of intersect_scene -- and the counters are the inline loop's; meshes are offered in the
// same order (and the offers are order-free anyway).
template <bool STATS>
__global__ void __launch_bounds__(BLOCK_THREADS, RT_MIN_WAVES) rt_wf_walk_kernel(const RenderArgs a) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t* stack = reinterpret_cast<uint32_t*>(lds_mem) + (threadIdx.x >> 6) * (stack_dwords(a) + a.tlas_entries * 64u) + lane;
    uint32_t* tstack = stack + stack_dwords(a);
    const LaneStack st{stack, a.stack_wide != 0u};
    const uint32_t n = *a.wf_count_in;
    const uint32_t tri0 = a.lay.tri_off;
    constexpr uint32_t PULL = 128u, NONE = 0xffffffffu;
    uint32_t pool_base = 0, pool_left = 0;  // wave-uniform: list entries reserved by this wave
    bool exhausted = (blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6)) * PULL >= n;  // (waves beyond the work never pull)
    enum : uint32_t { PH_IDLE = 0, PH_ADV = 1, PH_BOX = 2, PH_LEAF = 3 };
    uint32_t ph = PH_IDLE;
    bool offer = false;  // ADVANCE: the mesh just left has a hit to offer
    uint32_t slot = 0, it = 0, cur = 0, cur_count = 0, sp = 0, tsp = 0, mesh = 0, tnode = NONE;
    bool cull = false, cull_ok = false;
    f3 ro{0, 0, 0}, rd{0, 0, 1}, lo{0, 0, 0}, ld{0, 0, 0}, inv{0, 0, 0};
    Isect I;
    MeshBest b;
    b.t = INF;
    b.tri = NONE;
    b.u = b.v = 0.0f;
    int node_tests = 0, tri_tests = 0;

    // The record of the lane's next BOX step (a tree node's or a mesh node's two child boxes) is fetched the moment the
    // step is known -- at the end of the step before -- and used a pass later: the fetch latency (these are dependent
    // loads, L2 hits at best) runs under the other waves' passes instead of stalling this one's.
    // (ONE fetch site, at the end of the pass: the steps only mark the lane -- with a fetch in every branch that makes
    // a lane's next step a BOX step the compiler merged the branches' registers with copies, i.e. waited for the data
    // on the spot: 11.3 -> 13.3 ms per frame on the sponza-sized stand-in)
    float4 n0 = make_float4(0, 0, 0, 0), n1 = n0, n2 = n0, n3 = n0;
    bool want_fetch = false;
    auto fetch_box = [&]() { want_fetch = true; };
    // the tree iterator's step of intersect_scene: pop entries until one is a mesh (enter it) or a tree node (its
    // two boxes are the lane's next BOX step); an empty stack ends the item
    auto tree_pop = [&]() {
        if (tsp == 0u) {
            tnode = NONE;
            offer = false;
            ph = PH_ADV;
        } else {
            --tsp;
            const uint32_t e = tstack[tsp * 64u];
            if (e & 0x80000000u) {
                if (STATS) node_tests -= 2;  // counted with the tree item; the walk counts them again
                mesh = (e >> TLAS_REF_MESH_SHIFT) & TLAS_REF_MESH_MASK;
                cur = e & TLAS_REF_ROOT_MASK;
                cur_count = 0u;
                cull = (e & TLAS_REF_GLASS) == 0u;
                b.t = INF;
                sp = 0u;
                tnode = NONE;
            } else {
                tnode = e;
            }
            ph = PH_BOX;
            fetch_box();
        }
    };
    auto mesh_done = [&]() {
        if (b.tri != NONE) {
            offer = true;
            ph = PH_ADV;
        } else {
            tree_pop();
        }
    };
    auto pop_mesh = [&]() {
        if (sp == 0u) {
            mesh_done();
        } else {
            --sp;
            stack_get(st, sp, cur, cur_count);
            ph = cur_count ? PH_LEAF : PH_BOX;
            if (cur_count == 0u) fetch_box();
        }
    };

    for (;;) {
        const unsigned long long idle = __ballot(ph == PH_IDLE);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        // refill when a fair share of the wave is idle (a refill pass with one or two lanes costs what one with
        // sixteen does; 4 / 8 / 32 instead of 16 measured within 1 % of each other), or when nothing else is left to do
        if (n_idle != 0u && !exhausted && (n_idle >= 16u || n_idle == (uint32_t)__popcll(__ballot(true)))) {
            if (pool_left == 0u) {
                uint32_t t = 0;
                if (lane == 0u) t = atomicAdd(a.work_counter, PULL);
                t = __builtin_amdgcn_readfirstlane(t);
                if (t >= n) {
                    exhausted = true;
                } else {
                    pool_base = t;
                    pool_left = n - t < PULL ? n - t : PULL;
                }
            }
            if (pool_left != 0u) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                if (ph == PH_IDLE && rank < pool_left) {
                    DIAG(23);
                    slot = a.wf_list_in[pool_base + rank];
                    const float4* sq = wf_plane(a.wf_state, WF_STATE_PLANES, slot);
                    const float4 p1 = sq[1 * 64], p2 = sq[2 * 64];
                    ro = f3{p1.x, p1.y, p1.z};
                    rd = f3{p1.w, p2.x, p2.y};
                    I = Isect();
                    isect_spheres<false>(a, ro, rd, I);
                    it = 0u;
                    sp = tsp = 0u;
                    b.t = INF;
                    b.tri = NONE;
                    tnode = NONE;
                    offer = false;
                    ph = PH_ADV;
                }
                const uint32_t taken = n_idle < pool_left ? n_idle : pool_left;
                pool_base += taken;
                pool_left -= taken;
            }
        }
        const uint32_t nA = (uint32_t)__popcll(__ballot(ph == PH_ADV)), nB = (uint32_t)__popcll(__ballot(ph == PH_BOX)),
                       nL = (uint32_t)__popcll(__ballot(ph == PH_LEAF));
        if ((nA | nB | nL) == 0u) {
            if (exhausted) break;
            continue;
        }
        // Which kind of step runs: the most populated one.  (Tried instead, on the sponza-sized stand-in: the rare kinds --
        // leaf and bookkeeping steps, 6 and 4 of a ray's ~53 steps -- as soon as 8 / 16 / 24 / 32 / 40 lanes wait for
        // them, box steps otherwise: 14.3 / 12.6 / 12.6 / 14.7 / 16.5 ms per frame against 11.3 with this rule.)
        const uint32_t kind = (nB >= nA && nB >= nL) ? PH_BOX : nL >= nA ? PH_LEAF : PH_ADV;
        if (kind == PH_BOX) {
            if (ph == PH_BOX) {
                DIAG(20);
                const bool is_tree = tnode != NONE;
                const float4 q0 = n0, q1 = n1, q2 = n2, q3 = n3;  // (fetched when the step became known: fetch_box)
                const float tmax = is_tree ? INF : b.t;
                const float da = aabb_dist(lo, inv, q0, q1, tmax);
                const float db = aabb_dist(lo, inv, q2, q3, tmax);
                if (is_tree) {
                    DIAG(27);
                    const bool hit_a = !cull_ok || da < INF, hit_b = !cull_ok || db < INF;
                    if (hit_b) {
                        tstack[tsp * 64u] = fbits(q3.z) | (fbits(q3.w) ? 0x80000000u : 0u);
                        ++tsp;
                    }
                    if (hit_a) {
                        tstack[tsp * 64u] = fbits(q1.z) | (fbits(q1.w) ? 0x80000000u : 0u);
                        ++tsp;
                    }
                    tree_pop();
                } else {  // traverse_mesh's node visit
                    if (STATS) node_tests += 2;
                    const bool left_closer = da < db;
                    const float near_d = left_closer ? da : db, far_d = left_closer ? db : da;
                    const uint32_t near_i = fbits(left_closer ? q1.z : q3.z), near_c = fbits(left_closer ? q1.w : q3.w);
                    const uint32_t far_i = fbits(left_closer ? q3.z : q1.z), far_c = fbits(left_closer ? q3.w : q1.w);
                    if (far_d < b.t) {
                        stack_put(st, sp, far_i, far_c);
                        ++sp;
                    }
                    if (near_d < b.t) {
                        cur = near_i;
                        cur_count = near_c;
                        ph = cur_count ? PH_LEAF : PH_BOX;
                        if (cur_count == 0u) fetch_box();
                    } else {
                        pop_mesh();
                    }
                }
            }
        } else if (kind == PH_LEAF) {
            if (ph == PH_LEAF) {
                DIAG(21);
                if (STATS) tri_tests += (int)cur_count;
                for (uint32_t j = 0; j < cur_count; ++j) {
                    const uint32_t t = tri0 + (cur + j) * TRI_ISECT_BYTES;
                    tri_test<8>(lo, ld, ld4<false>(a, t), ld4<false>(a, t + 16), ld4<false>(a, t + 32), cull, cur + j, b);
                }
                pop_mesh();
            }
        } else {
            if (ph == PH_ADV) {
                DIAG(22);
                if (offer) {  // the mesh just left had a hit (wgsl:380-391)
                    DIAG(24);
                    f3 whp;
                    float wdst;
                    world_hit<false>(a, a.lay.mesh_off + mesh * MESH_REC_BYTES + 64u, lo, ld, ro, b.t, whp, wdst);
                    isect_offer(I, mesh, compact(b), whp, wdst);
                    b.tri = NONE;
                    offer = false;
                    tree_pop();
                }
                if (ph == PH_ADV) {  // the next item of the mesh loop (wgsl:369)
                    if (it >= a.n_items) {
                        DIAG(26);
                        wf_store_hit(a, slot, I);
                        ph = PH_IDLE;
                    } else {
                        DIAG(25);
                        const uint32_t io = a.lay.item_off + it * ITEM_BYTES;
                        const float4 item = ld4<false>(a, io);
                        const uint32_t kind = fbits(item.x), ia = fbits(item.y);
                        it += 1u;
                        if (kind & ITEM_NEW_XFORM) {
                            const uint32_t xo = a.lay.mesh_off + fbits(item.z) * MESH_REC_BYTES;
                            const float4 c0 = ld4<false>(a, xo), c1 = ld4<false>(a, xo + 16), c2 = ld4<false>(a, xo + 32), c3 = ld4<false>(a, xo + 48);
                            lo = mat_cols_xyz(c0, c1, c2, c3, ro, 1.0f);
                            ld = normalize3(mat_cols_xyz(c0, c1, c2, c3, rd, 0.0f));
                            inv = f3{rcp_(ld.x), rcp_(ld.y), rcp_(ld.z)};
                            cull_ok = rtm::abs_(inv.x) < INF && rtm::abs_(inv.y) < INF && rtm::abs_(inv.z) < INF &&
                                      rtm::abs_(lo.x) < INF && rtm::abs_(lo.y) < INF && rtm::abs_(lo.z) < INF;
                        }
                        if ((kind & ITEM_TLAS) == 0u) {
                            const float4 hdr = ld4<false>(a, io + 16);  // (flags, root, root count)
                            const uint32_t h_flags = fbits(hdr.x), h_root = fbits(hdr.y), h_count = fbits(hdr.z);
                            // root-box culling (intersect_scene's rule: internal roots only, finite rays only).  Written
                            // without a nested branch: the box is fetched and tested for every lane here and the result
                            // is only USED where the rule applies.  (With the test inside `if (cull_roots && count == 0)`
                            // hipcc 7.2 kept the OLD `cur` for the lanes that ran it -- visible in the ISA, found with
                            // rt_test_read_wavefront on a two-mesh scene.)
                            const uint32_t mo = a.lay.mesh_off + ia * MESH_REC_BYTES;
                            const float4 rmin = ld4<false>(a, mo + 160), rmax = ld4<false>(a, mo + 176);
                            const bool box_hit = aabb_dist(lo, inv, rmin, rmax, INF) < INF;
                            const bool tested = a.cull_roots != 0u && h_count == 0u && cull_ok;
                            const bool may_hit = !tested || box_hit;
                            if (STATS && !may_hit) node_tests += 2;  // the shader's two root-level tests (wgsl:322)
                            if (may_hit) {
                                mesh = ia;
                                cur = h_root;
                                cur_count = h_count;
                                cull = (h_flags & DMESH_GLASS) == 0u;
                                b.t = INF;
                                b.tri = NONE;
                                sp = 0u;
                                tnode = NONE;
                                ph = cur_count ? PH_LEAF : PH_BOX;
                                if (cur_count == 0u) fetch_box();
                            }
                        } else {
                            if (STATS) node_tests += 2 * (int)fbits(item.w);  // every mesh below counts its two root-level tests
                            tstack[0] = ia;
                            tsp = 1u;
                            tree_pop();
                        }
                    }
                }
            }
        }
        if (want_fetch) {  // the record of this lane's next BOX step
            const uint32_t wo = tnode != NONE ? a.lay.tlas_off + tnode * WIDE_REC_BYTES : a.lay.wide_off + cur * WIDE_REC_BYTES;
            n0 = ld4<false>(a, wo);
            n1 = ld4<false>(a, wo + 16);
            n2 = ld4<false>(a, wo + 32);
            n3 = ld4<false>(a, wo + 48);
            want_fetch = false;
        }
    }
    if (STATS && a.counters && __ballot((node_tests | tri_tests) != 0) != 0ull) {
        atomicAdd(&a.counters->node_tests, (unsigned long long)node_tests);
        atomicAdd(&a.counters->triangle_tests, (unsigned long long)tri_tests);
    }
}

