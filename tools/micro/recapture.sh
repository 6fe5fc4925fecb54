#!/bin/bash
# runs every mode of recapture_probe in its own process; a crash of one mode does not stop the others
D=$(dirname "$0")
for m in a b c d e; do
  timeout -k 5 60 "$D/recapture_probe" $m; echo "mode $m exit status $?"
done
exit 0
