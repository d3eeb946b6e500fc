cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3i; O=gpurun_out/r3i
python bench.py --campaign falcon9 > $O/campaign_falcon9.json 2> $O/campaign.err
python bench.py --campaign falcon9 > $O/campaign_falcon9_b.json 2>> $O/campaign.err
python -c "
import json, bench
print(json.dumps(bench.falcon9_leg(0)))" > $O/falcon9_leg.json 2> $O/leg.err
cat $O/campaign_falcon9.json $O/campaign_falcon9_b.json $O/falcon9_leg.json; tail -3 $O/leg.err
