/* The C ABI used from plain C (no C++, no torch, no HIP headers): include/instaorder_hip.h must compile as C, and the
 * planning entry points -- which never touch a device -- must work in a process that has no GPU at all.
 * Built and run by tests/test_host_cpu.py::test_header_is_plain_c_and_links. */
#include <stdio.h>
#include <string.h>

#include "instaorder_hip.h"

int main(void) {
    int heads[2] = {2, 3};
    io_net* net;
    io_tensor_info ti;
    long nparam, nrun;
    size_t ws_eval, ws_train;
    int i, ntens, convs = 0, bns = 0;

    if (io_abi_version() < 1) return 1;
    net = io_net_create(5, 2, heads);                 /* InstaOrderNet_od: resnet50_cls(in_channels=5, num_classes=[2,3]) */
    if (!net) { fprintf(stderr, "create: %s\n", io_last_error_string()); return 2; }
    nparam = io_net_param_floats(net);
    nrun = io_net_running_floats(net);
    ntens = io_net_num_tensors(net);
    for (i = 0; i < ntens; ++i) {
        if (io_net_tensor_info(net, i, &ti)) return 3;
        if (ti.kind == 0) ++convs;
        if (ti.kind == 1) ++bns;
    }
    ws_eval = io_net_workspace_bytes(net, 8, 256, 0);
    ws_train = io_net_workspace_bytes(net, 8, 256, 1);
    printf("tensors %d convs %d bns %d logits %d param_floats %ld running_floats %ld ws_eval %zu ws_train %zu dtype %d\n",
           ntens, convs, bns, io_net_num_logits(net), nparam, nrun, ws_eval, ws_train, io_net_get_dtype(net));
    if (io_net_set_dtype(net, IO_DTYPE_BF16)) return 4;
    printf("bf16 ws_train %zu\n", io_net_workspace_bytes(net, 8, 256, 1));
    /* an invalid request fails with a code and a message, it does not abort */
    if (io_net_workspace_bytes(net, 8, 250, 1) != 0 || strlen(io_last_error_string()) == 0) return 5;
    io_net_destroy(net);
    return 0;
}
