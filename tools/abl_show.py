import csv, glob, sys
for d in sorted(glob.glob('gpurun_out/abl_*/run_kernel_stats.csv')):
    print("==", d.split('/')[1])
    for r in csv.DictReader(open(d)):
        if any(k in r['Name'] for k in sys.argv[1:] or ['k_ppo_forward', 'k_ppo_backward', 'k_ppo_wgrad<']):
            print("   %-40s %4s %9.1f us" % (r['Name'][:40], r['Calls'], float(r['AverageNs']) / 1e3))
