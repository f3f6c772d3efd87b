for i in 1 2; do
PDGN_D_LATE=0 python3 tools/step_time.py 30 2>&1 | grep ms/step
PDGN_D_LATE=1 python3 tools/step_time.py 30 2>&1 | grep ms/step
done
