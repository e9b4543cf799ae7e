"""kpop_amd -- KPop's count -> twist -> distance hot path on AMD MI355X (gfx950).

The product is libkpop_hip.so (hand-written HIP kernels behind the C ABI of
include/kpop_hip.h); this package is the thin host-side mirror of the
reference's interface for that path.  Importing it does not touch the GPU;
`init(device)` does, and fails loudly when no GPU or no library is present.
"""
from .api import (check, COSINE, DNA_DS, DNA_SS, PROTEIN, EUCLIDEAN, METRIC_FLAT, METRIC_POWERS, MINKOWSKI, KPopError,  # noqa: F401
                  Twister, ca, count_reads, device_count, distance_rowwise, distance_summary, embeddings, init, summarize_distances,
                  metric_compute, parse_distance, splits_gaps, counter_stats, counter_combine, counter_transform, COMBINE_MEAN,
                  COMBINE_MEDIAN, TRANSF_BINARY, TRANSF_POWER, TRANSF_CLR, TRANSF_PSEUDO, Pipeline, host_empty, init_devices, use_device,
                  device_slots, OUT_TWISTED, OUT_DISTANCES, OUT_SUMMARY, Sharded, shard_bounds, sharded_distance_rowwise,
                  sharded_distance_summary)

__all__ = ["init", "device_count", "count_reads", "Twister", "ca", "metric_compute", "distance_rowwise",
           "distance_summary", "embeddings", "splits_gaps", "summarize_distances", "parse_distance", "KPopError", "DNA_DS", "DNA_SS", "PROTEIN", "EUCLIDEAN", "COSINE",
           "MINKOWSKI", "METRIC_FLAT", "METRIC_POWERS", "counter_stats", "counter_combine", "counter_transform",
           "COMBINE_MEAN", "COMBINE_MEDIAN", "TRANSF_BINARY", "TRANSF_POWER", "TRANSF_CLR", "TRANSF_PSEUDO", "Pipeline", "host_empty",
           "init_devices", "use_device", "device_slots", "OUT_TWISTED", "OUT_DISTANCES", "OUT_SUMMARY", "Sharded", "shard_bounds",
           "sharded_distance_rowwise", "sharded_distance_summary"]
