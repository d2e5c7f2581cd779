import json,sys
# usage: bench_summary.py [file]   (default: stdin)
src = open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin
for line in src:
    line=line.strip()
    if not line.startswith("{"): continue
    d=json.loads(line)
    print("value %.0f iter/s  round_us %.2f  pose_err %.2e" % (d["value"], d["roofline"]["launch_us"], d["pose_err_vs_gt"]))
    if "frame" in d: f=d["frame"]; print("frame %.1f fps match %.3f (full %.3f) join %.3f tr %.3f picp %.3f tri %.3f ms" % (f["frames_per_sec"], f["match_ms"], f.get("match_full_scan_ms", 0), f["join_ms"], f["transform_ms"], f["picp_ms"], f["triangulate_ms"]))
    if d.get("batched_frames"): b=d["batched_frames"]; print("batched frames: %.0f fps over %d GPU(s) (%d frames per GPU and call, %.1f us per frame per GPU)" % (b["frames_per_sec"], b["n_gpus"], b["frames_per_gpu"], b["us_per_frame_per_gpu"]))
    if "batched" in d: b=d["batched"]; print("batched %.3f ms (kernel %.3f)  %.0f iter/s  %.0f GB/s frac %.3f err %.2e" % (b["ms_per_call"], b.get("kernel_ms", 0), b["iters_per_sec"], b["roofline"]["achieved"], b["roofline"]["frac"], b["pose_err_vs_gt"]))
    if "cpu_baseline" in d: print("cpu %.0f iter/s" % d["cpu_baseline"]["value"])
    if "sequence" in d: q=d["sequence"]; print("sequence: %.0f fps (%d frames, %d..%d points, %d rounds) rmse_pos %.2e drift %.1e" % (q["frames_per_sec"], q["frames"], q["points_per_frame"]["min"], q["points_per_frame"]["max"], q["iters_per_frame"], q["accuracy_vs_ground_truth"]["rmse_position"], q["accuracy_vs_ground_truth"]["scale_ratio_drift"]))
    if "cpu_baseline" in d and "all_cores" in d["cpu_baseline"]: a=d["cpu_baseline"]["all_cores"]; print("cpu all cores: %.0f iter/s on %d threads" % (a["value"], a["cores"]))
