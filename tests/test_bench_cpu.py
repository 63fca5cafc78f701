"""Host-side pieces of bench.py that need no device."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_counter_means_of_a_rocprofv3_csv(tmp_path):
    """bench.measure_traffic's reader: per-family mean over the second half of a kernel's launches, only the requested
    counter, first matching family per kernel name."""
    sys.path.insert(0, ROOT)
    import bench
    rows = ["Correlation_Id,Kernel_Name,Counter_Name,Counter_Value"]
    bwd = "void ganet::(anonymous namespace)::layer_bwd_spec_kernel<false, true, 128>(long, float const*)"
    acc = "void ganet::(anonymous namespace)::layer_bwd_spec_kernel<true, true, 128>(long, float const*)"
    rb = "gsr::(anonymous namespace)::render_bwd_kernel(int, int, int)"
    for i, v in enumerate([900.0, 100.0, 200.0, 300.0]):            # second half: 200, 300
        rows.append(f'{i},"{bwd}",FETCH_SIZE,{v}')
        rows.append(f'{i},"{bwd}",WRITE_SIZE,{10 * v}')
    rows.append(f'9,"{acc}",FETCH_SIZE,777.0')                       # another variant of the family: not priced
    rows.append(f'10,"{rb}",FETCH_SIZE,42.0')
    p = tmp_path / "p_counter_collection.csv"
    p.write_text("\n".join(rows) + "\n")
    assert bench._counter_means(str(p), "FETCH_SIZE") == {"layer_bwd": 250.0, "render_bwd": 42.0}
    assert bench._counter_means(str(p), "WRITE_SIZE") == {"layer_bwd": 2500.0}


def test_config1_line_is_one_json_object():
    """`bench.py --config 1` (BASELINE.json configs[0], the reference's CPU-runnable case) needs no device and prints ONE
    JSON line with the contract's keys."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "1"], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "dtype", "data", "config",
              "cpu_baseline"):
        assert k in d, k
    assert d["cpu_baseline"]["kind"] in ("port", "reference") and d["cpu_baseline"]["cores"] >= 1
