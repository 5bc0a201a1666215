#!/bin/bash
# usage (GPU box): bash profiles/cli_init.sh <fasta>   -> where the device thread of the CLI spends its start-up, run back to back and with pauses
FA=$1
for pause in 0 0 0 1 1; do
  sleep $pause
  DPR_LOG=cli ./dipper_amd/bin/dipper -i m -I $FA -O /tmp/o.nwk -m 2 -d 2 2>&1 | grep -E "dpr_create|Input in|Main in" | tr '\n' '|'; echo " (pause $pause s)"
done
