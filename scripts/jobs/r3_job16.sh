./scripts/ubench/valu_rate.bin 2>&1 | grep -E "stream" 
