//! Batched variants of the four graph functions of `src/graph/mod.rs`, bound to the engine instead of the per-pair
//! `DistanceFn` (`src/graph/mod.rs:144-145`).  Same signatures apart from the distance argument, same results: the control
//! flow is the reference's, but every place that evaluated one pair at a time now asks the engine for the whole set of
//! pairs that step can need, in one call:
//!
//! * `search_layer`                 one `hx_distances_batch` group per popped candidate (its unvisited neighbours);
//! * `select_neighbors`             one `hx_pairwise` over the candidates; `check_element_closer` reads the table;
//! * `find_element_neighbors`       the two above;
//! * `update_neighbor_connections`  one `hx_pairwise_many` for all full neighbour lists of the new element at a layer.
//!
//! Element index == engine row id: `build_callback` appends a row with `hx_append_rows` exactly where it extends the
//! arena (`src/index/build.rs:451-454`) and drops it with `hx_pop_rows(1)` where it truncates (`:507-509`).
//! This is the *minimal* drop-in (one launch per expansion: correct, but latency-bound); the intended one hands whole
//! batches to `hx_index_insert` / `hx_index_search`, which run these same functions on the device.
use std::collections::{BinaryHeap, HashSet};

use super::ffi::*;
use crate::graph::{Candidate, ElementIdx, FurthestCandidate, GraphElement, NearestCandidate};
use crate::hnsw_constants::hnsw_get_layer_m;

/// The engine handle the build state owns instead of `dist_fmgr` (`src/index/build.rs:295-343`).
#[derive(Clone, Copy)]
pub struct GpuDistance {
    pub engine: *mut hx_engine,
}

impl GpuDistance {
    /// d(row `query`, row `ids[i]`) for every i, one launch.
    fn query_vs_rows(&self, query: ElementIdx, ids: &[u32]) -> Vec<f32> {
        let mut out = vec![0f32; ids.len()];
        if ids.is_empty() {
            return out;
        }
        let group_query = [query as u32];
        let offsets = [0u32, ids.len() as u32];
        unsafe {
            ck(
                self.engine,
                hx_distances_batch(self.engine, 1, group_query.as_ptr(), offsets.as_ptr(), ids.as_ptr(), out.as_mut_ptr()),
            );
        }
        out
    }

    /// Full w x w table among `ids` (row-major).
    fn pair_table(&self, ids: &[u32]) -> Vec<f32> {
        let w = ids.len();
        let mut out = vec![0f32; w * w];
        if w > 0 {
            unsafe { ck(self.engine, hx_pairwise(self.engine, ids.as_ptr(), w as u32, out.as_mut_ptr())) };
        }
        out
    }
}

/// `search_layer` (`src/graph/mod.rs:161-255`): the query is element `query` (its row is already in the engine).
pub fn search_layer_gpu(
    gpu: GpuDistance,
    elements: &[GraphElement],
    entry_points: &[Candidate],
    ef: usize,
    layer: i32,
    query: ElementIdx,
) -> Vec<Candidate> {
    let mut visited: HashSet<ElementIdx> = HashSet::with_capacity(ef * 2);
    let mut candidates: BinaryHeap<NearestCandidate> = BinaryHeap::new();
    let mut results: BinaryHeap<FurthestCandidate> = BinaryHeap::new();
    let mut result_len = 0usize;
    for ep in entry_points {
        visited.insert(ep.idx);
        candidates.push(NearestCandidate(*ep));
        results.push(FurthestCandidate(*ep));
        result_len += 1;
    }
    let mut batch: Vec<u32> = Vec::with_capacity(64);
    while let Some(NearestCandidate(c)) = candidates.pop() {
        let furthest = results.peek().map(|f| f.0.distance).unwrap_or(f32::MAX);
        if c.distance > furthest {
            break;
        }
        let c_elem = &elements[c.idx];
        if c_elem.level < layer {
            continue;
        }
        // the distances of one expansion do not depend on the heaps: gather the rows first, evaluate them in one launch,
        // then replay the reference's per-neighbour logic (mod.rs:226-243) over the returned slice, in list order
        batch.clear();
        for neighbor in &c_elem.neighbors[layer as usize].items {
            if !visited.insert(neighbor.idx) {
                continue;
            }
            if elements[neighbor.idx].level < layer {
                continue;
            }
            batch.push(neighbor.idx as u32);
        }
        let dist = gpu.query_vs_rows(query, &batch);
        for (&e, &e_distance) in batch.iter().zip(dist.iter()) {
            let always_add = result_len < ef;
            let furthest = results.peek().map(|f| f.0.distance).unwrap_or(f32::MAX);
            if e_distance < furthest || always_add {
                let cand = Candidate { distance: e_distance, idx: e as ElementIdx };
                candidates.push(NearestCandidate(cand));
                results.push(FurthestCandidate(cand));
                result_len += 1;
                if result_len > ef {
                    results.pop();
                    result_len -= 1;
                }
            }
        }
    }
    let mut out: Vec<Candidate> = results.into_iter().map(|f| f.0).collect();
    out.sort_by(|a, b| a.distance.partial_cmp(&b.distance).unwrap_or(std::cmp::Ordering::Equal));
    out
}

/// `select_neighbors` + `check_element_closer` (`src/graph/mod.rs:269-339`) on a precomputed pair table.
/// `table[i * w + j]` = d(candidates[i], candidates[j]).
fn select_on_table(candidates: &[Candidate], max_neighbors: usize, table: &[f32]) -> Vec<Candidate> {
    let w = candidates.len();
    if w <= max_neighbors {
        return candidates.to_vec();
    }
    let mut kept: Vec<usize> = Vec::with_capacity(max_neighbors); // indices into `candidates`
    let mut discarded: Vec<usize> = Vec::new();
    for i in 0..w {
        if kept.len() >= max_neighbors {
            break;
        }
        let closer = kept.iter().all(|&r| !(table[i * w + r] <= candidates[i].distance));
        if closer {
            kept.push(i);
        } else {
            discarded.push(i);
        }
    }
    for &d in &discarded {
        if kept.len() >= max_neighbors {
            break;
        }
        kept.push(d);
    }
    kept.into_iter().map(|i| candidates[i]).collect()
}

/// `select_neighbors` (`src/graph/mod.rs:269-308`), candidates sorted nearest first.
pub fn select_neighbors_gpu(gpu: GpuDistance, candidates: &[Candidate], max_neighbors: usize) -> Vec<Candidate> {
    if candidates.len() <= max_neighbors {
        return candidates.to_vec();
    }
    let ids: Vec<u32> = candidates.iter().map(|c| c.idx as u32).collect();
    let table = gpu.pair_table(&ids);
    select_on_table(candidates, max_neighbors, &table)
}

/// `find_element_neighbors` (`src/graph/mod.rs:355-427`).
pub fn find_element_neighbors_gpu(
    gpu: GpuDistance,
    elements: &mut [GraphElement],
    new_idx: ElementIdx,
    entry_idx: ElementIdx,
    ef_construction: usize,
    m: i32,
) {
    let new_level = elements[new_idx].level;
    let entry_level = elements[entry_idx].level;
    let d0 = gpu.query_vs_rows(new_idx, &[entry_idx as u32])[0];
    let mut ep = vec![Candidate { distance: d0, idx: entry_idx }];
    // greedy descent with ef = 1 down to new_level + 1
    let mut lc = entry_level;
    while lc >= new_level + 1 {
        let w = search_layer_gpu(gpu, elements, &ep, 1, lc, new_idx);
        if let Some(first) = w.first() {
            ep = vec![*first];
        }
        lc -= 1;
    }
    // ef_construction search + selection on every layer the new element lives in
    let mut lc = std::cmp::min(new_level, entry_level);
    while lc >= 0 {
        let lm = hnsw_get_layer_m(m, lc) as usize;
        let w = search_layer_gpu(gpu, elements, &ep, ef_construction, lc, new_idx);
        let selected = select_neighbors_gpu(gpu, &w, lm);
        elements[new_idx].neighbors[lc as usize].items = selected;
        ep = w; // the whole W seeds the next layer (mod.rs:425)
        lc -= 1;
    }
}

/// `update_neighbor_connections` (`src/graph/mod.rs:442-489`): all full lists of one layer pruned from ONE launch.
pub fn update_neighbor_connections_gpu(gpu: GpuDistance, elements: &mut [GraphElement], new_idx: ElementIdx, m: i32) {
    let new_level = elements[new_idx].level;
    for lc in (0..=new_level).rev() {
        let lm = hnsw_get_layer_m(m, lc) as usize;
        let snapshot: Vec<Candidate> = elements[new_idx].neighbors[lc as usize].items.clone();
        // lists with room take the back-link at once; the full ones become pair groups
        let mut full: Vec<(ElementIdx, Vec<Candidate>)> = Vec::new();
        for hc in &snapshot {
            let back = Candidate { distance: hc.distance, idx: new_idx };
            let list = &mut elements[hc.idx].neighbors[lc as usize];
            if list.len() < lm {
                list.items.push(back);
            } else {
                let mut all = list.items.clone();
                all.push(back);
                all.sort_by(|a, b| a.distance.partial_cmp(&b.distance).unwrap_or(std::cmp::Ordering::Equal));
                full.push((hc.idx, all));
            }
        }
        if full.is_empty() {
            continue;
        }
        // one lower-triangle group per full list (lm + 1 rows <= HX_PAIR_MAX_ROWS for m <= 31)
        let mut offsets: Vec<u32> = vec![0];
        let (mut na, mut nb): (Vec<u16>, Vec<u16>) = (Vec::new(), Vec::new());
        let mut ids: Vec<u32> = Vec::new();
        let mut out_offsets: Vec<u64> = Vec::new();
        let mut n_out: u64 = 0;
        for (_, all) in &full {
            let w = all.len();
            ids.extend(all.iter().map(|c| c.idx as u32));
            offsets.push(ids.len() as u32);
            na.push(w as u16);
            nb.push(0);
            out_offsets.push(n_out);
            n_out += (w * (w - 1) / 2) as u64;
        }
        let mut tri = vec![0f32; n_out as usize];
        unsafe {
            ck(
                gpu.engine,
                hx_pairwise_many(gpu.engine, full.len() as u32, offsets.as_ptr(), na.as_ptr(), nb.as_ptr(), ids.as_ptr(),
                                 out_offsets.as_ptr(), tri.as_mut_ptr()),
            );
        }
        for (g, (target, all)) in full.iter().enumerate() {
            let w = all.len();
            let t = &tri[out_offsets[g] as usize..];
            // expand the packed triangle (out[i*(i-1)/2 + j] = d(A_i, A_j), j < i) into the table select_on_table reads
            let mut table = vec![0f32; w * w];
            for i in 1..w {
                for j in 0..i {
                    let d = t[i * (i - 1) / 2 + j];
                    table[i * w + j] = d;
                    table[j * w + i] = d;
                }
            }
            elements[*target].neighbors[lc as usize].items = select_on_table(all, lm, &table);
        }
    }
}
