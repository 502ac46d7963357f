cd $GRAFT_REPO_ROOT
for n in 1 8; do python tools/rank_sim.py $n 390 2>&1 | tail -2; RANK_SIM_GRAPH=1 python tools/rank_sim.py $n 390 2>&1 | tail -2; done
