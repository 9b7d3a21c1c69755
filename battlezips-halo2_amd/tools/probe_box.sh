# diagnostic: run the default batch bench several times in one box session and log what varies between processes
for i in 1 2 3 4; do python - <<'PY'
import os, subprocess, sys, json, time
t0 = time.time()
p = subprocess.run([sys.executable, "bench.py", "--batch", "16", "--steps", "5", "--warmup", "3", "--no-cpu-baseline"], capture_output=True, text=True)
for l in p.stdout.splitlines():
    if l.startswith("{"):
        d = json.loads(l)
        print("value", round(d["value"], 1), "numa", d["config"].get("host_threads_pinned_to_numa_node"), "wall", round(time.time() - t0, 1), flush=True)
os.system("grep -E 'MemFree|MemAvailable|^Cached|AnonHugePages' /proc/meminfo | tr '\\n' ' '; echo")
os.system("cat /sys/fs/cgroup/memory.current /sys/fs/cgroup/memory.max 2>/dev/null | tr '\\n' ' '; echo")
PY
done > gpurun_out/runs.txt 2>&1
