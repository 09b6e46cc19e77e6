//! GPU distance backend for pgvector-rx's HNSW hot path (libhnswrx.so, MI355X).
pub mod ffi;
pub mod graph;
