"""Summarise rocprofv3 rocpd databases (kernel trace + PMC passes) into a text file for profiles/."""
import glob, os, sqlite3, sys

def q(db, sql):
    con = sqlite3.connect(db)
    try:
        cur = con.execute(sql)
        cols = [d[0] for d in cur.description]
        return cols, cur.fetchall()
    finally:
        con.close()

def main(prof_dir, out):
    lines = []
    tr = glob.glob(os.path.join(prof_dir, "trace", "*.db"))
    if tr:
        cols, rows = q(tr[0], "select name, count(*) as calls, sum(end-start) as total_ns, avg(end-start) as avg_ns, "
                              "min(end-start) as min_ns, max(end-start) as max_ns from kernels group by name order by total_ns desc")
        lines.append("== kernel trace (rocprofv3 --kernel-trace --stats) ==")
        lines.append("%-60s %6s %14s %14s %14s %14s" % ("kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms"))
        for r in rows:
            lines.append("%-60s %6d %14.3f %14.3f %14.3f %14.3f" % (r[0][:60], r[1], r[2]/1e6, r[3]/1e6, r[4]/1e6, r[5]/1e6))
        cols, rows = q(tr[0], "select name, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count "
                              "from kernels group by name") if True else (None, [])
        lines.append("")
        lines.append("%-60s %10s %6s %6s %8s %6s %6s %6s" % ("kernel", "grid", "wg", "lds", "scratch", "vgpr", "agpr", "sgpr"))
        for r in rows:
            lines.append("%-60s %10s %6s %6s %8s %6s %6s %6s" % ((r[0][:60],) + tuple(r[1:])))
    for sub in sorted(glob.glob(os.path.join(prof_dir, "pmc_*"))):
        if not os.path.isdir(sub):
            continue
        for db in glob.glob(os.path.join(sub, "*.db")):
            try:
                cols, rows = q(db, "select kernel_name, counter_name, count(*) as n, sum(value) as total, avg(value) as avg "
                                   "from counters_collection group by kernel_name, counter_name order by kernel_name, counter_name")
            except Exception as e:
                lines.append("(%s: %s)" % (db, e)); continue
            lines.append("")
            lines.append("== PMC pass %s (rocprofv3 --pmc; value summed over XCDs/SEs per dispatch, avg over dispatches) ==" % os.path.basename(sub))
            lines.append("%-50s %-24s %6s %22s" % ("kernel", "counter", "disp", "avg_per_dispatch"))
            for r in rows:
                lines.append("%-50s %-24s %6d %22.1f" % (r[0][:50], r[1], r[2], r[4]))
    txt = "\n".join(lines) + "\n"
    open(out, "w").write(txt)
    print(txt)
    update_traffic(prof_dir)
    update_valu_model(prof_dir)


def bench_key(bench):
    """bench.py's workload_key: the kernel is not part of the key (the entry names its kernels)."""
    c = bench["config"]
    K = c["K"]
    k = "%d-%d" % tuple(K) if isinstance(K, (list, tuple)) else str(K)
    return "events=%d|K=%s|reads=%s|iters=%d|chains=%d|paired=%d" % (
        c["events_per_gpu"], k, c["reads"], c["iters"], c["chains"], int("paired-end" in c["workload"])) + \
        ("|collapsed" if c.get("collapsed") else "")


def update_valu_model(prof_dir):
    """profiles/valu_model.json: what bench.py's VALU roofline is priced with.  Per profiled workload, from
    the rocprofv3 passes of this directory: VALU wave-instructions per chain-iteration (SQ_INSTS_VALU), the
    issue cycles they took while issuing (4 x SQ_ACTIVE_INST_VALU quad-cycles / SQ_INSTS_VALU), the share
    of the SIMDs' cycles that was (4 x SQ_ACTIVE_INST_VALU / (SIMDs x kernel cycles), kernel cycles =
    GRBM_GUI_ACTIVE / 8 XCDs) and the share of the wave slots that was occupied (SQ_WAVE_CYCLES)."""
    import json
    try:
        bench = json.loads(open(os.path.join(prof_dir, "bench_trace.json")).read().strip().split("\n")[-1])
        sdb = glob.glob(os.path.join(prof_dir, "pmc_sq", "*.db"))[0]
        fdb = glob.glob(os.path.join(prof_dir, "pmc_fetch", "*.db"))[0]
        tdb = glob.glob(os.path.join(prof_dir, "trace", "*.db"))[0]
    except (OSError, ValueError, IndexError):
        return
    def counters(db):
        _, rows = q(db, "select kernel_name, counter_name, avg(value) from counters_collection where kernel_name like "
                        "'%sampler_%' group by kernel_name, counter_name")
        out = {}
        for k, c, v in rows:
            out.setdefault(k, {})[c] = v
        return out
    sq, fe = counters(sdb), counters(fdb)
    _, rows = q(tdb, "select name, avg(end-start) from kernels where name like '%sampler_%' group by name")
    dur = {r[0]: r[1] for r in rows}
    # a launch of several kernels side by side: what the batch's HIP events bracket is first start -> last end
    _, iv = q(tdb, "select start, end from kernels where name like '%sampler_%' order by start")
    spans, cur = [], None
    for st, en in iv:
        if cur is None or st > cur[1]:
            if cur is not None:
                spans.append(cur[1] - cur[0])
            cur = [st, en]
        else:
            cur[1] = max(cur[1], en)
    if cur is not None:
        spans.append(cur[1] - cur[0])
    # (the first launch of a batch may carry trial runs; the median launch is the one the bench times)
    span_ns = sorted(spans)[(len(spans) - 1) // 2] if spans else None
    c = bench["config"]
    chain_iters = float(c["events_per_gpu"]) * c["chains"] * (c["iters"] + 1)
    kernels = {}
    for k, v in sq.items():
        if "SQ_INSTS_VALU" not in v:
            continue
        cycles = fe.get(k, {}).get("GRBM_GUI_ACTIVE", 0.0) / 8.0          # per XCD
        kernels[k] = {
            "valu_instructions": v["SQ_INSTS_VALU"], "active_valu_quadcycles": v["SQ_ACTIVE_INST_VALU"],
            "wave_quadcycles": v.get("SQ_WAVE_CYCLES"), "waves": v.get("SQ_WAVES"),
            "kernel_ns": dur.get(k), "kernel_cycles": cycles,
            "issue_cycles_per_valu": 4.0 * v["SQ_ACTIVE_INST_VALU"] / v["SQ_INSTS_VALU"],
            "valu_busy": (4.0 * v["SQ_ACTIVE_INST_VALU"] / (1024.0 * cycles)) if cycles else None,
            "wave_slot_occupancy": (4.0 * v["SQ_WAVE_CYCLES"] / (2048.0 * cycles)) if cycles and v.get("SQ_WAVE_CYCLES") else None,
        }
    if not kernels:
        return
    tot_valu = sum(k["valu_instructions"] for k in kernels.values())
    tot_act = sum(k["active_valu_quadcycles"] for k in kernels.values())
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "valu_model.json")
    try:
        table = json.load(open(path))
    except (OSError, ValueError):
        table = {}
    table[bench_key(bench)] = {
        "source": os.path.basename(prof_dir.rstrip("/")), "chain_iterations": chain_iters,
        "valu_per_chain_iteration": tot_valu / chain_iters,
        "issue_cycles_per_valu": 4.0 * tot_act / tot_valu,
        "kernels": kernels,
        # the traced launch as bench.py's HIP events see it, and the shader clock the library's probe measured in that very
        # (kernel-trace) pass: bench.py compares shader CYCLES of its run with these
        # (the launch as the bench's own HIP events timed it in that pass -- the very quantity a later run is compared with;
        # the trace's first-start-to-last-end span of the median launch where the line carries none)
        "launch_span_ns": (bench.get("roofline") or {}).get("kernel_ms") and bench["roofline"]["kernel_ms"] * 1e6 or span_ns,
        "trace_span_ns": span_ns,
        "clock_ghz": (bench.get("roofline") or {}).get("clock_ghz"),
    }
    json.dump(table, open(path, "w"), indent=1, sort_keys=True)
    print("valu model:", bench_key(bench), {k: round(v, 3) if isinstance(v, float) else v
                                            for k, v in table[bench_key(bench)].items() if k != "kernels"})


def update_traffic(prof_dir):
    """profiles/traffic.json: HBM bytes per launch of the sampler kernel of this profiled bench run.
    FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (gfx950 counts 64 B per 128 B request,
    MI355X_MICROARCH.md section HBM); WRITE_SIZE is taken as reported (uncalibrated there)."""
    import json
    try:
        bench = json.loads(open(os.path.join(prof_dir, "bench_trace.json")).read().strip().split("\n")[-1])
        fdb = glob.glob(os.path.join(prof_dir, "pmc_fetch", "*.db"))[0]
        wdb = glob.glob(os.path.join(prof_dir, "pmc_write", "*.db"))[0]
    except (OSError, ValueError, IndexError):
        return
    def avg(db, counter):
        _, rows = q(db, "select kernel_name, avg(value) from counters_collection where counter_name='%s' "
                        "and kernel_name like '%%sampler_%%' group by kernel_name order by avg(value) desc" % counter)
        return rows[0][1] if rows else None
    fetch_kb, write_kb = avg(fdb, "FETCH_SIZE"), avg(wdb, "WRITE_SIZE")
    if fetch_kb is None or write_kb is None:
        return
    key = bench_key(bench)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    try:
        table = json.load(open(path))
    except (OSError, ValueError):
        table = {}
    table[key] = {"fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
                  "hbm_bytes_per_launch": 2 * fetch_kb * 1024 + write_kb * 1024,
                  "source": os.path.basename(prof_dir.rstrip("/")),
                  # (the compact bench line carries the rate; bytes = rate x the run's kernel time)
                  "algorithmic_bytes_per_launch": bench["roofline"].get("algorithmic_bytes_per_launch",
                      bench["roofline"]["algorithmic_GBs"] * 1e9 * bench["roofline"]["kernel_ms"] * 1e-3)}
    json.dump(table, open(path, "w"), indent=1, sort_keys=True)
    print("traffic:", key, table[key])

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
