const char *vivit_hip_source_hash(void) { return "23b55f42271adc8582fe3cbd1040fc53-262e1d34"; }
