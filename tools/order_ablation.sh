#!/bin/bash
# Dev aid: what direction ordering costs and buys (run on the GPU box from the repo root).
A="$@ --steps 400 --warmup 40 --no-cpu-baseline --no-extras"
pick() { python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('%-44s %7.2f us/step  kernel %6.2f us' % (sys.argv[1], 1e3*d['ms_per_step'], d['roofline']['kernel_us']))" "$1"; }
python3 bench.py $A | pick "no order, random directions"
python3 bench.py $A --presorted-directions | pick "runs flag, presorted directions"
python3 bench.py $A --direction-order | pick "order, random directions"
python3 bench.py $A --direction-order --presorted-directions | pick "order, presorted directions"
