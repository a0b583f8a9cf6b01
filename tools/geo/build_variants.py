"""Build experiment variants of the fused kernels WITHOUT build switches in the product sources: every variant is a set of
textual patches applied to a scratch copy of envidr_amd/csrc (tools/geo/variants/<name>/src), whose patched translation unit
is compiled and linked with the product's other objects into tools/geo/variants/<name>.so (selected at run time through
ENVIDR_AMD_LIB).  A patch whose `old` text is not found exactly once is an error, so variants cannot silently rot.

    python tools/geo/build_variants.py [name ...]

Earlier forms of the kernels that these switches used to select (the 64-samples-per-wave k_geo_eval, Jacobian parked in global
scratch, LDS-shared weight stream of the persistent kernel, un-piped split layers, section timers) were removed from the
sources in round 3; they are in the history up to commit 42365ab.
"""
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from envidr_amd import build as B  # noqa: E402

# name -> (translation unit, [(file, old, new), ...])
P1 = "        // ================= phase 1: this lane's eight levels: values + Jacobian -> LDS ============================\n"
P2 = '        f32x16 o3, gf[1];\n'
P3 = '            wp.template end_pass<kSdfFrags>();\n        }\n\n        // ================= phase 3'
VARIANTS = {
    "base": ("geometry_pass", []),
    "geo_ahead1": ("geometry_pass", [("geometry_pass.hip", "constexpr int kGeoAhead = 2;", "constexpr int kGeoAhead = 1;")]),
    "geo_ahead3": ("geometry_pass", [("geometry_pass.hip", "constexpr int kGeoAhead = 2;", "constexpr int kGeoAhead = 3;")]),
    "geo_ring4": ("geometry_pass", [("geometry_pass.hip", "constexpr int kGeoRing = 8;", "constexpr int kGeoRing = 4;")]),
    "e16_waves16": ("geometry_pass", [("geo_eval16.hip.h", "constexpr int kE16Waves = 12;", "constexpr int kE16Waves = 16;")]),
    "e16_waves8": ("geometry_pass", [("geo_eval16.hip.h", "constexpr int kE16Waves = 12;", "constexpr int kE16Waves = 8;")]),
    # chunk prediction off: every live ray takes the round's cap (its count at most doubles per round)
    "no_chunk_prediction": ("geometry_pass", [("geometry_pass.hip", "a.predict = (r >= 1 && r + 1 < rounds) ? 1u : 0u;", "a.predict = 0u;"),
                                              ("geometry_pass.hip", "chunks[r] = r == 0 ? std::min(16u, d->max_steps) : d->max_steps;",
                                               "chunks[r] = r == 0 ? std::min(16u, d->max_steps) : (r < 6 ? std::min(d->max_steps, 8u << r) : d->max_steps);")]),
    # per-section clocks of the per-ray rounds -> counters[80 + 16 round + 2 section] (sum over blocks, in 64-tick units) and
    # [... + 1] (max over blocks): sections 0 prologue + count records, 1 composite, 2 first hit / counting march, 3 allocate + write
    # march, 4 state write-back, 5 finished rays' outputs; words 12 / 13 of a round: whole block sum / max, word 14: blocks
    # (read back with tools/geo/rays_timers_probe.py)
    "rays_timers": ("geometry_pass", [
        ("geometry_pass.hip", "        bool alive = false;          // still needs samples after this round's compositing\n",
         "        bool alive = false;\n        unsigned long long tm_[7]; tm_[0] = tm_[1] = tm_[2] = __builtin_amdgcn_s_memtime();\n"),
        ("geometry_pass.hip", "            const uint32_t wave_records = wave_sum(k);\n",
         "            tm_[1] = __builtin_amdgcn_s_memtime();\n            const uint32_t wave_records = wave_sum(k);\n"),
        ("geometry_pass.hip", "            if (active) {\n                finish = terminated || last || st.n_taken >= a.max_samples;\n",
         "            __builtin_amdgcn_s_waitcnt(0); tm_[2] = __builtin_amdgcn_s_memtime();\n"
         "            if (active) {\n                finish = terminated || last || st.n_taken >= a.max_samples;\n"),
        ("geometry_pass.hip", "        const uint32_t wave_slots = wave_sum(want);\n",
         "        tm_[3] = __builtin_amdgcn_s_memtime();\n        const uint32_t wave_slots = wave_sum(want);\n"),
        ("geometry_pass.hip", "        if (alive) {\n            st.chunk_count = marched | ((marched < chunk) ? 0x80000000u : 0u);\n",
         "        tm_[4] = __builtin_amdgcn_s_memtime();\n        if (alive) {\n            st.chunk_count = marched | ((marched < chunk) ? 0x80000000u : 0u);\n"),
        ("geometry_pass.hip", "        if (finish) {\n            // run_cuda epilogue for this ray",
         "        tm_[5] = __builtin_amdgcn_s_memtime();\n        if (finish) {\n            // run_cuda epilogue for this ray"),
        ("geometry_pass.hip", "            if (a.ray_cost) a.ray_cost[id] = (uint16_t)min(st.n_taken, 65535u);\n        }\n",
         "            if (a.ray_cost) a.ray_cost[id] = (uint16_t)min(st.n_taken, 65535u);\n        }\n"
         "        tm_[6] = __builtin_amdgcn_s_memtime();\n"
         "        if (lane == 0) {\n"
         "            uint32_t* w_ = a.counters + 80u + 16u * min(a.round, 7u);\n"
         "            for (int s_ = 0; s_ < 6; ++s_) { const uint32_t d_ = (uint32_t)((tm_[s_ + 1] - tm_[s_]) >> 6); atomicAdd(w_ + 2 * s_, d_); atomicMax(w_ + 2 * s_ + 1, d_); }\n"
         "            const uint32_t all_ = (uint32_t)((tm_[6] - tm_[0]) >> 6); atomicAdd(w_ + 12, all_); atomicMax(w_ + 13, all_); atomicAdd(w_ + 14, 1u);\n"
         "        }\n"),
    ]),
    "grid_full": ("geometry_pass", [("geometry_pass.hip", "const dim3 grid(r == 0 ? ray_blocks : std::min(ray_blocks, 1024u));", "const dim3 grid(ray_blocks);")]),
    "tcache16": ("geometry_pass", [("geometry_pass.hip", "constexpr uint32_t kTimeCache = 32;", "constexpr uint32_t kTimeCache = 16;")]),
    "grid_full_tcache16": ("geometry_pass", [("geometry_pass.hip", "const dim3 grid(r == 0 ? ray_blocks : std::min(ray_blocks, 1024u));", "const dim3 grid(ray_blocks);"),
                                             ("geometry_pass.hip", "constexpr uint32_t kTimeCache = 32;", "constexpr uint32_t kTimeCache = 16;")]),
    # ablations of k_geo_eval32 (what bounds it?): no matrix phase / no table gathers / neither
    "eval_nomlp": ("geometry_pass", [("geometry_pass.hip", "            pipe_layer_from_lanes<kLevels, 2, kSdfW1, kSdfN>(wp, lane, in, h1);\n            pipe_layer_from_tiles<2, 2, kSdfW2, kSdfN, true>(wp, lane, h1, h2);\n            pipe_layer16_from_tiles<2, kSdfW3, kSdfN, true>(wp, lane, h2, o3);        // 64 -> 15 on 16-row MFMA blocks (k_order 2)\n",
                                      "            for (int r_ = 0; r_ < 16; ++r_) { h1[0][r_] = in[r_]; h1[1][r_] = in[r_]; h2[0][r_] = in[r_]; h2[1][r_] = in[r_]; o3[r_] = in[r_]; }\n"),
                                     ("geometry_pass.hip", "            pipe_layer_from_tiles<2, 2, kSdfW2t, kSdfN, false, false>(wp, lane, g2, g1);\n", "            g1[0] = g2[1]; g1[1] = g2[0];\n"),
                                     ("geometry_pass.hip", "            pipe_layer_from_tiles<2, 1, kSdfW1t, kSdfN, false, false>(wp, lane, g1, gf);\n            wp.template end_pass<kSdfFrags>();\n", "            gf[0] = g1[0] + g1[1];\n")]),
    "eval_nogather": ("geometry_pass", [("hash_lean.hip.h", "        st.row[2 * j] = __builtin_amdgcn_raw_buffer_load_b64(table, r0[j] * 8u + (LANE_LEVEL ? lv.row0_bytes : 0u), LANE_LEVEL ? 0u : lv.row0_bytes, AUX);\n        st.row[2 * j + 1] = __builtin_amdgcn_raw_buffer_load_b64(table, r1[j] * 8u + (LANE_LEVEL ? lv.row0_bytes : 0u), LANE_LEVEL ? 0u : lv.row0_bytes, AUX);\n",
                                         "        st.row[2 * j] = u32x2{r0[j], lv.row0_bytes};\n        st.row[2 * j + 1] = u32x2{r1[j], lv.row0_bytes};\n")]),
    # ablations of the record-shading kernel (timing only: results are wrong): where does a round's time go besides the MFMAs?
    "shade_noide": ("fused_render", [("fused_render.hip", "        ide_eval<IDE_DEG, true>(vx, vy, vz, kinv, [&](int j, float re, float im) {\n            code[j] = re * c.light_scale;\n            code[TERMS + j] = im * c.light_scale;\n        });\n",
                                      "        for (int j = 0; j < TERMS; ++j) { code[j] = vx * kinv + (float)j; code[TERMS + j] = vy * vz - (float)j; }\n")]),
    "shade_noenv": ("fused_render", [("fused_render.hip", "            env_pass<TERMS, ENV_T, kEnvN>(wp, lane, aux, in, o);                        // env_pass.hip.h\n",
                                      "            for (int r_ = 0; r_ < 16; ++r_) o[r_] = in[r_] + in[r_ + 16];\n")]),
    "shade_noheads": ("fused_render", [("fused_render.hip", "            pipe_layer_from_lanes<kDSteps, 1, kHeadD1, kHeadN>(wp, lane, in_d, d1);\n            pipe_layer16_from_tiles<1, kHeadD2, kHeadN, true>(wp, lane, d1, d2);\n            pipe_layer_from_lanes<kSSteps, 2, kHeadS1, kHeadN>(wp, lane, in_s, s1);\n            pipe_layer_from_tiles<2, 2, kHeadS2, kHeadN, true>(wp, lane, s1, s2);\n            pipe_layer16_from_tiles<2, kHeadS3, kHeadN, true>(wp, lane, s2, s3);\n            wp.template end_pass<kHeadFrags>();\n",
                                        "            for (int r_ = 0; r_ < 16; ++r_) { d2[r_] = in_d[r_ % kDSteps]; s3[r_] = in_s[r_ % kSSteps]; }\n")]),
    # per-wave start / end clocks of the record-shading kernel -> the unused tail of c_diffuse (tools/probe/wave_times.py reads them):
    # do all waves take the same time for the same number of rounds?
    "shade_wavetime": ("fused_render", [
        ("fused_render.hip", "    const uint32_t waves = gridDim.x * (kBlockThreads / 64);\n    // a frame whose records did not fit",
         "    const unsigned long long t_start_ = __builtin_amdgcn_s_memrealtime();\n    const unsigned long long c_start_ = __builtin_amdgcn_s_memtime();\n    const uint32_t waves = gridDim.x * (kBlockThreads / 64);\n    // a frame whose records did not fit"),
        ("fused_render.hip", "        if (on) {\n#pragma unroll\n            for (int d = 0; d < 3; ++d) { a.c_diffuse[3 * i + d] = cd[d]; a.c_specular[3 * i + d] = cs[d]; }\n        }\n    }\n}\n",
         "        if (on) {\n#pragma unroll\n            for (int d = 0; d < 3; ++d) { a.c_diffuse[3 * i + d] = cd[d]; a.c_specular[3 * i + d] = cs[d]; }\n        }\n    }\n"
         "    if (lane == 0) { unsigned long long* o_ = reinterpret_cast<unsigned long long*>(a.c_diffuse + 3 * (size_t)(a.M - 8192)) + 4 * (size_t)blockIdx.x;\n"
         "        o_[0] = t_start_; o_[1] = __builtin_amdgcn_s_memrealtime(); o_[2] = __builtin_amdgcn_s_memtime() - c_start_; o_[3] = __builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 8); }\n}\n"),
    ]),
    "shade_noload": ("fused_render", [
        ("fused_render.hip", "        const size_t gi = a.slot ? (size_t)a.slot[i] : i;          // where this record's geometry lives\n        float nrm[3], vd[3], geo[12];\n        const size_t ray = a.ray_ids ? (size_t)a.ray_ids[i] : 0;\n",
         "        const size_t gi = i;\n        float nrm[3], vd[3], geo[12];\n        const size_t ray = 0;\n"),
        ("fused_render.hip", "        for (int d = 0; d < 3; ++d) { nrm[d] = on ? a.normals[3 * gi + d] : 0.0f; vd[d] = on ? dir[d] : 0.0f; }\n#pragma unroll\n        for (int j = 0; j < 12; ++j) geo[j] = a.geo_feat[(size_t)a.geo_stride * gi + j];\n        const float rough = a.roughness[(size_t)a.rough_stride * gi];\n",
         "        for (int d = 0; d < 3; ++d) { nrm[d] = 0.577f + 1e-9f * (float)(gi + d); vd[d] = -0.577f + 1e-9f * (float)(gi & 255); }\n#pragma unroll\n        for (int j = 0; j < 12; ++j) geo[j] = 0.288f + 1e-9f * (float)(gi + j);\n        const float rough = 0.1f + 1e-9f * (float)(gi & 1023); (void)dir;\n")]),
    # ablations / parameters of k_env_split (timing only where marked: results are wrong)
    "split_base": ("shade_split", []),
    "split_nodma": ("shade_split", [("mlp_split.hip.h", "            if constexpr (c == kSplitMeetAt + 1) dma_piece<0>();\n            if constexpr (c == kSplitMeetAt + 3) dma_piece<1>();\n", "")]),       # timing only
    "split_halfdma": ("shade_split", [("mlp_split.hip.h", "            if constexpr (c == kSplitMeetAt + 3) dma_piece<1>();\n", "")]),       # timing only
    "split_nobarrier": ("shade_split", [("mlp_split.hip.h", 'asm volatile("s_waitcnt vmcnt(%0)\\n\\ts_barrier" ::"n"(kSplitPieces * (kSplitSlots - 4)) : "memory");', "")]),       # timing only (racy)
    "split_nt": ("shade_split", [("mlp_split.hip.h", "offen lds\\n", "offen nt lds\\n")]),
    "split_sc": ("shade_split", [("mlp_split.hip.h", "offen lds\\n", "offen sc1 lds\\n")]),
    # L2 channel hot-spot experiment: workgroup i streams copy i % 4 of the blob, copies 512 B + a blob apart (run with
    # tools/geo/split_probe.py --copies 4, which tiles the blob accordingly)
    "split_copies4": ("shade_split", [("shade_split.hip", "    wp.start(s_w, lane, wave, blob, L::Padded);",
                                       "    wp.start(s_w, lane, wave, (const char*)blob + (blockIdx.x & 3u) * (((uint32_t)L::Padded * 1024u + kSplitBlobGrain - 1u) / kSplitBlobGrain * kSplitBlobGrain + 512u), L::Padded);")]),
    "split_nowait": ("shade_split", [("mlp_split.hip.h", 'asm volatile("s_waitcnt vmcnt(%0)\\n\\ts_barrier" ::"n"(kSplitPieces * (kSplitSlots - 4)) : "memory");', 'asm volatile("s_barrier" ::: "memory");')]),       # timing only (racy)
    "split_res8": ("shade_split", [("mlp_split.hip.h", "constexpr int kSplitResident = 88; ", "constexpr int kSplitResident = 8;  ")]),
    "split_res48": ("shade_split", [("mlp_split.hip.h", "constexpr int kSplitResident = 88; ", "constexpr int kSplitResident = 48; ")]),
    "split_sc01nt": ("shade_split", [("mlp_split.hip.h", "offen lds\\n", "offen sc0 sc1 nt lds\\n")]),
    "split_sc0": ("shade_split", [("mlp_split.hip.h", "offen lds\\n", "offen sc0 lds\\n")]),
    "split_noconvert": ("shade_split", [("mlp_split.hip.h", "    split_f16(a, hi, lo); h[r % 8] = hi; l[r % 8] = lo;\n    split_f16(b, hi, lo); h[(r + 1) % 8] = hi; l[(r + 1) % 8] = lo;\n",
                                          "    hi = (_Float16)a; lo = (_Float16)b; h[r % 8] = hi; l[r % 8] = lo; h[(r + 1) % 8] = lo; l[(r + 1) % 8] = hi;\n")]),       # timing only
    # matrix-phase token per SIMD: of the waves that share a SIMD only one is in the MFMA phase at a time
    "eval_token": ("geometry_pass", [
        ("geometry_pass.hip", "    __shared__ float s_jac[kE32Waves * kE32Steps * 6 * 64];\n    const uint32_t lane = lane_id();\n",
         "    __shared__ float s_jac[kE32Waves * kE32Steps * 6 * 64];\n    __shared__ uint32_t s_token[4];\n    const uint32_t lane = lane_id();\n    const uint32_t simd = __builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11));\n"),
        ("geometry_pass.hip", "        if (threadIdx.x < kLevels) s_lv[threadIdx.x] = a.lv[threadIdx.x];\n    }\n    __syncthreads();\n    WeightLdsRing<kGeoRing> wp;",
         "        if (threadIdx.x < kLevels) s_lv[threadIdx.x] = a.lv[threadIdx.x];\n        if (threadIdx.x < 4) s_token[threadIdx.x] = 0;\n    }\n    __syncthreads();\n    WeightLdsRing<kGeoRing> wp;"),
        ("geometry_pass.hip", "            f32x16 h1[2], h2[2];\n            pipe_layer_from_lanes<kLevels, 2, kSdfW1, kSdfN>(wp, lane, in, h1);\n",
         "            f32x16 h1[2], h2[2];\n            if (lane == 0) while (__hip_atomic_exchange(&s_token[simd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) __builtin_amdgcn_s_sleep(2);\n            __builtin_amdgcn_sched_barrier(0);\n            pipe_layer_from_lanes<kLevels, 2, kSdfW1, kSdfN>(wp, lane, in, h1);\n"),
        ("geometry_pass.hip", "            pipe_layer_from_tiles<2, 1, kSdfW1t, kSdfN, false, false>(wp, lane, g1, gf);\n            wp.template end_pass<kSdfFrags>();\n",
         "            pipe_layer_from_tiles<2, 1, kSdfW1t, kSdfN, false, false>(wp, lane, g1, gf);\n            wp.template end_pass<kSdfFrags>();\n            __builtin_amdgcn_sched_barrier(0);\n            if (lane == 0) __hip_atomic_store(&s_token[simd], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n"),
    ]),
    # how much do the empty rounds of a hinted frame cost?  Nothing measurable: 38.27 ms against 38.26 - 38.38 with seven (three rounds make
    # the un-hinted warm-up frame's last round hand out so much that it overflows its buffers)
    "rounds4": ("geometry_pass", [("geometry_pass.hip", "    constexpr uint32_t kRounds = 7;", "    constexpr uint32_t kRounds = 4;")]),
    "first_chunk12": ("geometry_pass", [("geometry_pass.hip", "std::min(16u, d->max_steps)", "std::min(12u, d->max_steps)")]),
    "first_chunk24": ("geometry_pass", [("geometry_pass.hip", "std::min(16u, d->max_steps)", "std::min(24u, d->max_steps)")]),
    # ---- the fused-pair split kernel (round 6), timing only unless noted ----
    "s2_base": ("shade_split2", []),
    # the round's prologue (record loads, IDE, layer-1 operands -> LDS) only in a workgroup's first round
    "s2_noprologue": ("shade_split2", [("shade_split2.hip", "        ide_eval<IDE_DEG, true>(enc ? wr[0]", "        if (base == blockIdx.x * 128u) ide_eval<IDE_DEG, true>(enc ? wr[0]")]),
    "s2_nobarrier": ("shade_split2", [("mlp_split2.hip.h", 'asm volatile("s_waitcnt vmcnt(%0)\\n\\ts_barrier" ::"n"(kS2Slots - 3) : "memory");', "")]),       # racy
    "s2_nodma": ("shade_split2", [("mlp_split2.hip.h", "        dma(fill_lds, fill_off);                          // the piece aimed at the last barrier\n", "")]),
    "s2_ahead2": ("shade_split2", [("mlp_split2.hip.h", "constexpr int kS2Ahead = 4; ", "constexpr int kS2Ahead = 2; "),
                                   ("mlp_split2.hip.h", "kS2ChunkFrags == 8 && kS2Ahead == 4 && kS2MeetAt == 4", "kS2ChunkFrags == 8 && kS2MeetAt == 4")]),
    "s2_floor": ("shade_split2", [("mlp_split2.hip.h", 'asm volatile("s_waitcnt vmcnt(%0)\\n\\ts_barrier" ::"n"(kS2Slots - 3) : "memory");', ""),
                                  ("mlp_split2.hip.h", "        dma(fill_lds, fill_off);                          // the piece aimed at the last barrier\n", ""),
                                  ("shade_split2.hip", "        ide_eval<IDE_DEG, true>(enc ? wr[0]", "        if (base == blockIdx.x * 128u) ide_eval<IDE_DEG, true>(enc ? wr[0]")]),
    "s2_noconvert": ("shade_split2", [("mlp_split.hip.h", "    split_f16(a, hi, lo); h[r % 8] = hi; l[r % 8] = lo;\n    split_f16(b, hi, lo); h[(r + 1) % 8] = hi; l[(r + 1) % 8] = lo;\n",
                                       "    hi = (_Float16)a; lo = (_Float16)b; h[r % 8] = hi; l[r % 8] = lo; h[(r + 1) % 8] = lo; l[(r + 1) % 8] = hi;\n")]),
    "split_group4": ("shade_split", [("mlp_split.hip.h", "constexpr int kSplitGroup = 2;", "constexpr int kSplitGroup = 4;")]),
    "ring16": ("fused_render", [("fused_render.hip", "constexpr int kRingDepth = 32;", "constexpr int kRingDepth = 16;")]),
    # issue priority by phase (MI355X_MICROARCH.md "Two waves per SIMD": VALU issue is arbitrated by priority, then age): does the SIMD
    # partner's hash arithmetic advance under this wave's MFMAs when the hash phase outranks the matrix phase?
    "eval_prio_hash": ("geometry_pass", [("geometry_pass.hip", P1, "        __builtin_amdgcn_s_setprio(2);\n" + P1),
                                         ("geometry_pass.hip", P2, "        __builtin_amdgcn_s_setprio(0);\n" + P2)]),
    "eval_prio_hash3": ("geometry_pass", [("geometry_pass.hip", P1, "        __builtin_amdgcn_s_setprio(2);\n" + P1),
                                          ("geometry_pass.hip", P2, "        __builtin_amdgcn_s_setprio(0);\n" + P2),
                                          ("geometry_pass.hip", P3, "            wp.template end_pass<kSdfFrags>();\n            __builtin_amdgcn_s_setprio(2);\n        }\n\n        // ================= phase 3")]),
    "eval_prio_mfma": ("geometry_pass", [("geometry_pass.hip", P1, "        __builtin_amdgcn_s_setprio(0);\n" + P1),
                                         ("geometry_pass.hip", P2, "        __builtin_amdgcn_s_setprio(2);\n" + P2)]),
    # per-phase clocks of the range owners (sum over waves, in 64-cycle units) -> the first words of the mask scratch (tools/probe/scatter_only.py --timers)
    "scatter_timers": ("hashencoder", [
        ("hashencoder.hip", "            for (uint32_t r0 = 0; r0 < most; r0 += 64u) {\n                Batch q;\n",
         "            for (uint32_t r0 = 0; r0 < most; r0 += 64u) {\n                Batch q;\n                const unsigned long long T0 = __builtin_amdgcn_s_memtime();\n"),
        ("hashencoder.hip", "                issue_loads(q);\n                Prepared prep[kLdsBatch];\n",
         "                const unsigned long long T1 = __builtin_amdgcn_s_memtime();\n                issue_loads(q);\n                __builtin_amdgcn_s_waitcnt(0);\n                const unsigned long long T2 = __builtin_amdgcn_s_memtime();\n                Prepared prep[kLdsBatch];\n"),
        ("hashencoder.hip", "                for (uint32_t u = 0; u < kLdsBatch; ++u) add(prep[u]);\n            }\n",
         "                for (uint32_t u = 0; u < kLdsBatch; ++u) if (prep[u].pending == 0xffffffffu) T_acc[3] += 1;\n                __builtin_amdgcn_sched_barrier(0);\n                const unsigned long long T3 = __builtin_amdgcn_s_memtime();\n"
         "#pragma unroll\n                for (uint32_t u = 0; u < kLdsBatch; ++u) add(prep[u]);\n                __builtin_amdgcn_s_waitcnt(0);\n                const unsigned long long T4 = __builtin_amdgcn_s_memtime();\n"
         "                T_acc[0] += T1 - T0; T_acc[1] += T2 - T1; T_acc[2] += T3 - T2; T_acc[3] += T4 - T3;\n            }\n"),
        ("hashencoder.hip", "        uint4 m_next[kLdsBatch];\n", "        unsigned long long T_acc[4] = {0, 0, 0, 0};\n        const unsigned long long T_begin = __builtin_amdgcn_s_memtime();\n        uint4 m_next[kLdsBatch];\n"),
        ("hashencoder.hip", "        __syncthreads();\n        float* t = grad_table + ((size_t)row0 + base) * C;",
         "        if (lane == 0) { for (int q_ = 0; q_ < 4; ++q_) atomicAdd(timers + q_, (unsigned long long)(T_acc[q_] >> 6)); atomicAdd(timers + 4, (unsigned long long)((__builtin_amdgcn_s_memtime() - T_begin) >> 6)); }\n"
         "        __syncthreads();\n        float* t = grad_table + ((size_t)row0 + base) * C;"),
        ("hashencoder.hip", "    constexpr uint32_t kRows = kLdsScatterFloats / C;\n    __shared__ float s_acc[kLdsScatterFloats];\n",
         "    constexpr uint32_t kRows = kLdsScatterFloats / C;\n    __shared__ float s_acc[kLdsScatterFloats];\n    unsigned long long* timers = reinterpret_cast<unsigned long long*>(grad_table + (size_t)offsets[L] * C);\n"),
    ]),
    # block-level phases of the range owners (thread 0 of every workgroup, 64-cycle units): zeroing, point loop (barrier to barrier), flush, whole kernel
    "scatter_phases": ("hashencoder", [
        ("hashencoder.hip", "    constexpr uint32_t kRows = kLdsScatterFloats / C;\n    __shared__ float s_acc[kLdsScatterFloats];\n",
         "    constexpr uint32_t kRows = kLdsScatterFloats / C;\n    __shared__ float s_acc[kLdsScatterFloats];\n    unsigned long long* timers = reinterpret_cast<unsigned long long*>(grad_table + (size_t)offsets[L] * C);\n    const unsigned long long K0 = __builtin_amdgcn_s_memtime();\n    unsigned long long P_acc[3] = {0, 0, 0};\n"),
        ("hashencoder.hip", "        const uint32_t base = range * kRows;\n        for (uint32_t i = threadIdx.x; i < kLdsScatterFloats; i += kLdsScatterThreads) s_acc[i] = 0.0f;\n        __syncthreads();\n",
         "        const uint32_t base = range * kRows;\n        const unsigned long long Q0 = __builtin_amdgcn_s_memtime();\n        for (uint32_t i = threadIdx.x; i < kLdsScatterFloats; i += kLdsScatterThreads) s_acc[i] = 0.0f;\n        __syncthreads();\n        const unsigned long long Q1 = __builtin_amdgcn_s_memtime();\n"),
        ("hashencoder.hip", "        __syncthreads();\n        float* t = grad_table + ((size_t)row0 + base) * C;",
         "        __syncthreads();\n        const unsigned long long Q2 = __builtin_amdgcn_s_memtime();\n        float* t = grad_table + ((size_t)row0 + base) * C;"),
        ("hashencoder.hip", '        asm volatile("s_waitcnt lgkmcnt(0)\\n\\ts_barrier" ::: "memory");\n    }\n',
         '        asm volatile("s_waitcnt lgkmcnt(0)\\n\\ts_barrier" ::: "memory");\n        P_acc[0] += Q1 - Q0; P_acc[1] += Q2 - Q1; P_acc[2] += __builtin_amdgcn_s_memtime() - Q2;\n    }\n'),
        ("hashencoder.hip", "    if (level >= L) break;                                   // past the last super-group (uniform over the workgroup)\n",
         "    if (level >= L) { if (threadIdx.x == 0) { for (int q_ = 0; q_ < 3; ++q_) atomicAdd(timers + q_, (unsigned long long)(P_acc[q_] >> 6)); const unsigned long long all_ = (__builtin_amdgcn_s_memtime() - K0) >> 6; atomicAdd(timers + 3, all_); atomicMax(timers + 4, all_); } break; }\n"),
    ]),
    "scatter_waves8": ("hashencoder", [("hashencoder.hip", "constexpr uint32_t kLdsScatterThreads = 1024;", "constexpr uint32_t kLdsScatterThreads = 512;")]),
    "scatter_batch1": ("hashencoder", [("hashencoder.hip", "constexpr uint32_t kLdsBatch = 2; ", "constexpr uint32_t kLdsBatch = 1; ")]),
    "scatter_batch4": ("hashencoder", [("hashencoder.hip", "constexpr uint32_t kLdsBatch = 2; ", "constexpr uint32_t kLdsBatch = 4; ")]),
    "scatter_parts192": ("hashencoder", [("hashencoder.hip", "constexpr uint32_t kLdsBlocksPerLevel = 288;", "constexpr uint32_t kLdsBlocksPerLevel = 192;")]),
    "scatter_parts288": ("hashencoder", [("hashencoder.hip", "constexpr uint32_t kLdsBlocksPerLevel = 288;", "constexpr uint32_t kLdsBlocksPerLevel = 288;")]),
    "scatter_parts512": ("hashencoder", [("hashencoder.hip", "constexpr uint32_t kLdsBlocksPerLevel = 288;", "constexpr uint32_t kLdsBlocksPerLevel = 512;")]),
    "scatter_parts384": ("hashencoder", [("hashencoder.hip", "constexpr uint32_t kLdsBlocksPerLevel = 288;", "constexpr uint32_t kLdsBlocksPerLevel = 384;")]),
    # timing only: the LDS atomics as plain LDS writes / as nothing
    "scatter_plainwrite": ("hashencoder", [("hashencoder.hip", "                    for (int c = 0; c < C; ++c) __hip_atomic_fetch_add(&s_acc[c * kRows + at], v[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n",
                                            "                    for (int c = 0; c < C; ++c) s_acc[c * kRows + at] = v[c];\n")]),
    "scatter_noatomic": ("hashencoder", [("hashencoder.hip", "                    for (int c = 0; c < C; ++c) __hip_atomic_fetch_add(&s_acc[c * kRows + at], v[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n",
                                          "                    for (int c = 0; c < C; ++c) if (v[c] == 123.456f) s_acc[c * kRows + at] = v[c];\n")]),
}



VARIANTS["scatter_timers_noatomic"] = ("hashencoder", VARIANTS["scatter_timers"][1] + VARIANTS["scatter_noatomic"][1])


def main(names):
    B.build(verbose=False)
    out = ROOT / "tools" / "geo" / "variants"
    out.mkdir(exist_ok=True)
    for name in names or VARIANTS:
        unit, patches = VARIANTS[name]
        # same depth below the repository root as envidr_amd/csrc: the sources include "../../include/*.h"
        src = out / name / "csrc"
        if src.parent.exists():
            shutil.rmtree(src.parent)
        shutil.copytree(B.CSRC, src, ignore=shutil.ignore_patterns("build", "*.o"))
        inc = out / "include"
        if inc.is_symlink() or inc.exists():
            inc.unlink()
        inc.symlink_to(ROOT / "include")
        for fname, old, new in patches:
            text = (src / fname).read_text()
            if text.count(old) != 1:
                raise SystemExit(f"variant {name}: patch text not found exactly once in {fname}: {old!r}")
            (src / fname).write_text(text.replace(old, new))
        obj = out / f"{name}.o"
        cmd = [B.hipcc(), *B.HIPCC_FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", str(src / f"{unit}.hip"), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            print(name, "FAILED\n", r.stderr[-3000:])
            continue
        info = [line.split("remark:")[1].strip().replace("[-Rpass-analysis=kernel-resource-usage]", "") for line in r.stderr.splitlines()
                if any(k in line for k in ("VGPRs:", "VGPRs Spill", "ScratchSize", "LDS Size"))]
        objs = [str(o) for o in sorted((B.CSRC / "build").glob("*.o")) if o.stem != unit]
        lib = out / f"{name}.so"
        r = subprocess.run([B.hipcc(), f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-fno-gpu-rdc", *objs, str(obj), "-o", str(lib)],
                           capture_output=True, text=True)
        print(name, "|", " ".join(info)[:600], "|", "link ok" if r.returncode == 0 else r.stderr[-500:])


if __name__ == "__main__":
    main(sys.argv[1:])
