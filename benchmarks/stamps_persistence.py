"""Is a block's scan time a property of the block?  From the raw per-block stamps of benchmarks/fused_stamps.py
(VDB_STAMPS_RAW=file): correlation of the blocks' scan times between consecutive launches (each block's time relative to
its group's median), whether block b runs on the same CU every launch, and how much of the spread a per-block share
would remove (spread of the per-block MEAN over launches against the spread inside one launch)."""
import json, sys
import numpy as np
raw = json.load(open(sys.argv[1]))
dur = np.array([r["dur_us"] for r in raw])            # [reps][grid]
hw = np.array([r["hw_id"] for r in raw])
xcc = np.array([r["xcc"] for r in raw])
reps, grid = dur.shape
n_cu = 256
grp = (np.arange(grid) & 7) + 8 * np.minimum(np.arange(grid) // n_cu, 3)
rel = np.empty_like(dur)
for r in range(reps):
    for g in np.unique(grp):
        m = grp == g
        rel[r, m] = dur[r, m] / np.median(dur[r, m]) - 1.0
corr = [float(np.corrcoef(rel[r], rel[r + 1])[0, 1]) for r in range(reps - 1)]
same_cu = float(np.mean(hw[1:] == hw[:-1]))
same_xcc = float(np.mean(xcc[1:] == xcc[:-1]))
mean_rel = rel.mean(axis=0)
print(json.dumps({"reps": reps, "grid": grid,
                  "corr_of_relative_block_time_between_consecutive_launches[min,median,max]": [min(corr), float(np.median(corr)), max(corr)],
                  "block_on_same_hw_id_as_in_the_previous_launch": same_cu, "same_xcc": same_xcc,
                  "std_of_relative_time_inside_a_launch": float(rel.std(axis=1).mean()),
                  "std_of_per_block_mean_over_launches": float(mean_rel.std()),
                  "std_left_after_removing_per_block_mean": float((rel - mean_rel).std()),
                  "max_relative_inside_a_launch_median": float(np.median(rel.max(axis=1))),
                  "max_after_removing_per_block_mean_median": float(np.median((rel - mean_rel).max(axis=1)))}))
