module lattigo_ring_hip

go 1.21
