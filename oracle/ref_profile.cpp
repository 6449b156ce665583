// placeholder until the profile harness is written
int main() { return 0; }
