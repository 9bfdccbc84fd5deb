#!/bin/bash
# Diagnostic: the GPU test-suite under rocgdb, native backtraces if the process faults.  bash tools/gdb_pytest.sh <log>
LOG=${1:-gpurun_out/gdb_pytest.log}
timeout -k 10 1000 /opt/rocm/bin/rocgdb -q -batch -ex "set pagination off" -ex "set confirm off" -ex "handle SIGSEGV stop print nopass" \
  -ex "handle SIG33 SIG34 SIG35 SIGUSR1 SIGUSR2 SIGPIPE SIGCHLD nostop noprint pass" -ex run -ex "bt 60" -ex "info threads" -ex "thread apply all bt 15" \
  --args python -m pytest tests -x -q -m gpu > "$LOG" 2>&1
echo "rocgdb exit $?" >> "$LOG"
