// rt_api_hybrid_blob.inl -- a STATEMENT FRAGMENT of rt_upload_scene (rt_api.hip includes it inside the function, with
// -DRT_EXPERIMENTS=1 only): the small blob of the hybrid launches of a deferred-walk sequence (option "hybrid"; DESIGN.md 5.4:
// built, parity-tested, measured no faster) -- the scene without the deferred mesh's BVH and triangles.  Possible when that
// mesh's wide records and triangles are the LAST of their arrays (indices are absolute: the other meshes' then form a prefix) --
// true of scenes that add one big model to a small set.  Fills `small`, `sl`, `small_need`, `small_ok`.  Moved out of
// rt_api.hip in round 5 as it was (code motion).
        if (have_defer) {
            const uint32_t d = defer_mesh;
            bool last = wide_base[d] + defer_internal == (uint32_t)wide.size();
            for (uint32_t i = 0; i < n_meshes; ++i)
                if (i != d) {
                    if (tri_hi[i] > tri_lo[d] || wide_base[i] > wide_base[d]) last = false;
                    small_need = std::max(small_need, mesh_need[i]);
                }
            if (last && tri_lo[d] <= n_triangles) {
                const uint32_t nw = wide_base[d], nt = tri_lo[d];
                uint64_t o = 0;
                sl.mesh_off = (uint32_t)o;   o += (uint64_t)n_meshes * MESH_REC_BYTES;
                sl.mat_off = (uint32_t)o;    o += (uint64_t)(n_meshes + n_spheres) * MATERIAL_BYTES;
                sl.sphere_off = (uint32_t)o; o += (uint64_t)n_spheres * SPHERE_BYTES;
                sl.item_off = (uint32_t)o;   o += (uint64_t)items.size() * ITEM_BYTES;
                sl.tlas_off = (uint32_t)o;   o += (uint64_t)tlas.size() * WIDE_REC_BYTES;
                sl.forest_off = (uint32_t)o; o += (uint64_t)forest_entries.size() * FOREST_ENTRY_BYTES;
                sl.wide_off = (uint32_t)o;   o += (uint64_t)nw * WIDE_REC_BYTES;
                sl.tri_off = (uint32_t)o;    o += (uint64_t)nt * TRI_ISECT_BYTES;
                sl.shade_off = (uint32_t)o;  o += (uint64_t)nt * TRI_SHADE_BYTES;
                sl.bytes = (uint32_t)o;
                if (sl.mat_off != lay.mat_off || sl.wide_off != lay.wide_off) o = (uint64_t)LDS_BUDGET_BYTES + 1;  // (cannot happen: same leading sections)
                if (o <= LDS_BUDGET_BYTES) {
                    small.assign(o / 16, make_float4(0, 0, 0, 0));
                    auto copy = [&](uint32_t dst, uint32_t src, uint64_t bytes) {
                        if (bytes) memcpy(small.data() + dst / 16, blob.data() + src / 16, bytes);
                    };
                    copy(sl.mesh_off, lay.mesh_off, (uint64_t)n_meshes * MESH_REC_BYTES);
                    copy(sl.wide_off, lay.wide_off, (uint64_t)nw * WIDE_REC_BYTES);
                    copy(sl.tri_off, lay.tri_off, (uint64_t)nt * TRI_ISECT_BYTES);
                    copy(sl.shade_off, lay.shade_off, (uint64_t)nt * TRI_SHADE_BYTES);
                    copy(sl.mat_off, lay.mat_off, (uint64_t)(n_meshes + n_spheres) * MATERIAL_BYTES);
                    copy(sl.sphere_off, lay.sphere_off, (uint64_t)n_spheres * SPHERE_BYTES);
                    copy(sl.item_off, lay.item_off, (uint64_t)items.size() * ITEM_BYTES);
                    copy(sl.forest_off, lay.forest_off, (uint64_t)forest_entries.size() * FOREST_ENTRY_BYTES);
                    small_ok = true;
                }
            }
        }
