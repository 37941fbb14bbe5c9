// tad_arima.hip — placeholder until the ARIMA(1,1,1) kernel lands (next milestone).
#include "tad_internal.h"
namespace tad {
size_t arima_workspace_bytes(Grid) { return 0; }
int launch_arima(hipStream_t, Grid, const double *, const uint32_t *, int, double *, DevCounters *, void *, size_t) { return -1; }
}  // namespace tad
