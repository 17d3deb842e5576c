"""The ONE line bench.py prints last: the driver parses the final stdout line, and a line that carries every leg does not survive its
tail buffer (round 5: 20.9 KB -> `parsed: null`).  Everything the legs measured goes to a side file (bench_legs.json) and to earlier
stdout lines; the headline keeps the contract's keys plus `roofline`, `rank_roofline`, `cpu_baseline`, `parity` and a few figures of
the end-to-end legs, and stays under MAX_BYTES.

Pure functions over dicts: tests/test_bench_headline.py runs them on canned legs without a GPU."""
import json

MAX_BYTES = 4096
RANK_TARGET_FRAC = 0.60              # BASELINE.json north_star: ">= 60 % of HBM roofline for FM-index rank"

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline", "parity")

# the result type first, the lane width second: scores are the reference's int32 values; the kernel runs the recurrence in 16-bit lanes only
# for jobs whose scheme / band / length bound proves that exact (roofline.a32 is the all-int32 kernel on the same launch)
DTYPE = "int32 results (int16 lanes where proven exact; all-int32 kernel in roofline.a32)"


def _num(v, digits=6):
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    if isinstance(v, float):
        return float("%.*g" % (digits, v))
    return v


def _pick(d, keys):
    """the named keys of a dict that are present and scalar (numbers rounded)"""
    if not isinstance(d, dict):
        return None
    return {k: _num(d[k]) for k in keys if k in d and not isinstance(d[k], (dict, list))}


def compact_roofline(r):
    if not isinstance(r, dict):
        return None
    out = _pick(r, ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "gcups", "hbm_GBs", "hbm_frac"))
    c = r.get("counters")
    if isinstance(c, dict):
        out["counters"] = _pick(c, ("l2_hit_rate", "lds_bank_conflict_frac", "lds_bank_conflict_cycles"))
    e = r.get("executed")
    if isinstance(e, dict):
        out["executed"] = _pick(e, ("valu_lane_ops_per_cell", "frac"))
    a32 = r.get("a32")
    if isinstance(a32, dict):
        out["a32"] = _pick(a32, ("kernel_ms", "reads_per_s", "frac", "identical_results"))
    return out


def compact_rank(r):
    if not isinstance(r, dict):
        return None
    out = _pick(r, ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "queries", "random_lines_per_s_G", "random_line_limit_G"))
    out["target"] = RANK_TARGET_FRAC
    out["met"] = bool(isinstance(r.get("frac"), (int, float)) and r["frac"] >= RANK_TARGET_FRAC)
    return out


def compact_cpu(c):
    if not isinstance(c, dict):
        return None
    out = _pick(c, ("value", "unit", "cores", "kind", "host_physical_cores"))
    s = c.get("sample")
    if isinstance(s, str):
        out["sample"] = s[:200]
    g = c.get("gpu_vs_cpu_on_sample")
    if isinstance(g, dict):
        out["bit_exact_vs_gpu"] = g.get("bit_exact")
    return out


def compact_e2e(e):
    """a few figures of the end-to-end legs (config 4 / config 5); the whole objects are in the legs file"""
    if not isinstance(e, dict):
        return None
    out = {}
    c4 = e.get("config4_full_size")
    if isinstance(c4, dict):
        out["config4_Mreads_per_s"] = _num(c4.get("Mreads_per_s"))
        out["config4_reads"] = c4.get("reads")
        d = c4.get("default")
        if isinstance(d, dict):
            out["config4_default"] = _pick(d, ("Mreads_per_s", "identical_to_reference_layout"))
    rr = e.get("repeat_rich")
    if isinstance(rr, dict):
        out["repeat_rich"] = _pick(rr, ("Mreads_per_s", "reads", "batch_reads", "rounds", "extensions", "identical_to_reference_layout"))
    c5 = e.get("config5_per_gpu_share")
    if isinstance(c5, dict):
        out["config5_share"] = {k: _num((c5.get(k) or {}).get("Mpairs_per_s")) for k in ("serial", "two_batches_in_flight") if isinstance(c5.get(k), dict)}
    p = e.get("parity")
    if isinstance(p, dict):
        out["parity_bit_exact"] = p.get("bit_exact")
    return out or None


def headline(full, legs_file=None):
    """full: the dict bench.py assembled (contract keys + every leg) -> the compact dict of the final line"""
    h = {}
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline"):
        h[k] = _num(full.get(k), 9)
    h["dtype"] = DTYPE
    h["data"] = full.get("data", "synthetic")
    cfg = full.get("config") or {}
    h["config"] = {k: cfg[k] for k in ("workload", "reads_per_gpu", "read_len", "band", "type", "parallelism", "gather", "gather_path", "rccl_ranks_seen", "per_rank_ms_per_step")
                   if k in cfg and cfg[k] is not None}
    if isinstance(h["config"].get("workload"), str):
        h["config"]["workload"] = h["config"]["workload"][:240]
    h["roofline"] = compact_roofline(full.get("roofline"))
    if full.get("rank_roofline") is not None:
        h["rank_roofline"] = compact_rank(full.get("rank_roofline"))
    h["cpu_baseline"] = compact_cpu(full.get("cpu_baseline"))
    h["parity"] = full.get("parity")
    e2e = compact_e2e(full.get("e2e_leg"))
    if e2e:
        h["e2e"] = e2e
    sh = full.get("e2e_sharded_leg")
    if isinstance(sh, dict):
        h["e2e_sharded"] = _pick(sh, ("Mreads_per_s", "reads", "ms_per_batch", "gather_path", "error"))
    if legs_file:
        h["legs_file"] = legs_file
    return h


def headline_line(full, legs_file=None):
    """-> the final stdout line (no newline); raises if it would not fit or would not parse"""
    h = headline(full, legs_file)
    line = json.dumps(h, separators=(",", ":"))
    if len(line.encode()) >= MAX_BYTES:
        # shed the optional parts, the largest first, rather than emit a line the driver cannot parse
        for k in ("e2e", "e2e_sharded", "rank_roofline"):
            if k in h and len(line.encode()) >= MAX_BYTES:
                h[k] = {"see": legs_file or "legs"}
                line = json.dumps(h, separators=(",", ":"))
    if len(line.encode()) >= MAX_BYTES:
        raise ValueError("headline line is %d bytes (limit %d)" % (len(line.encode()), MAX_BYTES))
    json.loads(line)
    return line


def leg_lines(full):
    """one stdout line per leg, printed BEFORE the headline: {"leg": name, "data": {...}}"""
    contract = set(REQUIRED) | {"rank_roofline"}
    return [json.dumps({"leg": k, "data": v}) for k, v in full.items() if k not in contract or k in ("roofline", "rank_roofline", "cpu_baseline")]
