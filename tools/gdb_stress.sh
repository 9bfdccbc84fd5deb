#!/bin/bash
# Diagnostic: the in-process torch.cuda users of the GPU suite, several pytest sessions in ONE process, under rocgdb.
LOG=${1:-gpurun_out/gdb_stress.log}
cat > /tmp/stress_sessions.py <<'P'
import sys, pytest
for k in range(int(sys.argv[1])):
    rc = pytest.main(["tests/test_gpu_frame.py", "-x", "-q", "-m", "gpu", "-k", "strips or sharded or output_target or phases or moving or textures or texture"])
    print("session", k, "rc", rc, flush=True)
    if rc != 0:
        sys.exit(1)
P
timeout -k 10 1000 /opt/rocm/bin/rocgdb -q -batch -ex "set pagination off" -ex "set confirm off" -ex "set print thread-events off" \
  -ex "handle SIGSEGV stop print nopass" -ex "handle SIG33 SIG34 SIG35 SIGUSR1 SIGUSR2 SIGPIPE SIGCHLD nostop noprint pass" -ex run -ex "bt 60" \
  -ex "info threads" -ex "thread apply all bt 12" --args python /tmp/stress_sessions.py ${2:-8} > "$LOG" 2>&1
echo "rocgdb exit $?" >> "$LOG"
