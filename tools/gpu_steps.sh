#!/bin/bash
# Runs the given steps ("name|timeout_s|command") one after the other on the GPU box; each under its own timeout, output to gpurun_out/<name>.log.
# A step that fails goes on to the next; a step that TIMES OUT or is killed (status >= 124) stops the chain (nothing else touches the GPU).
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for s in "$@"; do
  name=${s%%|*}; rest=${s#*|}; to=${rest%%|*}; cmd=${rest#*|}
  echo "[step $name] start $(date +%T)"
  timeout -k 10 $to bash -c "$cmd" > gpurun_out/$name.log 2>&1
  rc=$?
  echo "[step $name] exit $rc $(date +%T)"; tail -3 gpurun_out/$name.log
  if [ $rc -ge 124 ]; then echo "[step $name] timed out or was killed: stopping"; exit $rc; fi
done
exit 0
