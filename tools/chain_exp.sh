for e in 0 1 2 3 4 7; do echo "EXP=$e"; E4S_CHAIN_EXP=$e timeout 120 python tools/time_chain.py 2>&1 | grep "chain" | cut -c1-150; done
