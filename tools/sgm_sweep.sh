# experiment: frames per group x pipelining of groups (env knobs of an experimental build)
for pp in 0 1; do for g in 4 6 8 12; do f=24; echo "pipe=$pp group=$g frames=$f"; SGM_F=$f MOD_SGM_PIPE=$pp MOD_SGM_GROUP=$g timeout -k 10 100 python tools/time_sgm.py 2>/dev/null | grep "paths="; done; done
