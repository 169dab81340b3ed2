// launch_tiles.js — one Node process per GPU, from JavaScript (the reference's host language): the multi-GPU launcher of
// the JavaScript host, counterpart of `python -m torch.distributed.run … bench.py` on the Python side.
//
//   const { launchTiles } = require('./launch_tiles.js');
//   const results = await launchTiles({ gpus: 8, tilesX: 1, halo: 56, worker: '/abs/path/worker.js', args: {...} });
//
// The parent forks `gpus` children; child `rank` drives HIP device `devices[rank]` (default: its rank; every process sees all
// GPUs, the arrangement RCCL is exercised with most — `isolate: true` hides the others through HIP_VISIBLE_DEVICES).  Rank 0
// creates the communicator id (ncclGetUniqueId inside libfluid_hip.so) and hands it to the parent, which passes it on to
// the other ranks; every child then builds its tile with createFluid({ tile: { rank, world, tilesX, halo, commId } }) —
// a collective ncclCommInitRank — and runs the worker module's exported function `(sim, ctx) => result` on it, where
// ctx = { rank, world, tilesX, args }.  From there sim.step() exchanges ghost rows / columns with its neighbours by
// ncclSend / ncclRecv inside the library; nothing else crosses between the processes.  Results come back in rank order.
// The reference itself is a single-context page (script.js has no counterpart); see INTEGRATION.md §4.
'use strict';
const { fork } = require('child_process');
const path = require('path');

function launchTiles (options) {
    const world = options.gpus || 1;
    const tilesX = options.tilesX || 1;
    if (world % tilesX) throw new Error('launchTiles: gpus must be a multiple of tilesX');
    const devices = options.devices || Array.from({ length: world }, (_, i) => i);
    return new Promise((resolve, reject) => {
        const children = [], results = new Array(world);
        let done = 0, failed = false;
        const fail = err => {
            if (failed) return;
            failed = true;
            children.forEach(c => { try { c.kill(); } catch (e) { /* already gone */ } });
            reject(err);
        };
        for (let rank = 0; rank < world; rank++) {
            const env = Object.assign({}, process.env, { HSA_ENABLE_IPC_MODE_LEGACY: '0' });   // dmabuf IPC: what RCCL across processes needs here
            if (options.isolate) env.HIP_VISIBLE_DEVICES = String(devices[rank]);
            const child = fork(__filename, ['--tile-child'], { env, stdio: ['ignore', 'inherit', 'inherit', 'ipc'] });
            children.push(child);
            child.on('message', m => {
                if (m.type === 'commId') {            // from rank 0: pass the id to everyone (rank 0 included: it waits for it too)
                    children.forEach(c => c.send({ type: 'commId', id: m.id }));
                } else if (m.type === 'result') {
                    results[rank] = m.value;
                    if (++done === world) resolve(results);
                } else if (m.type === 'error') {
                    fail(new Error('rank ' + rank + ': ' + m.message));
                }
            });
            child.on('exit', code => { if (code !== 0 && results[rank] === undefined) fail(new Error('rank ' + rank + ' exited with code ' + code)); });
            // linkModel: [latency us, GB/s], or 'calibrate' (the default on more than one GPU): every rank measures its links behind commInit
            const linkModel = options.linkModel !== undefined ? options.linkModel : (world > 1 ? 'calibrate' : undefined);
            child.send({ type: 'init', rank, world, tilesX, device: options.isolate ? 0 : devices[rank], halo: options.halo === undefined ? 56 : options.halo, reach: options.reach, linkModel,
                         worker: path.resolve(options.worker), fluid: options.fluid || {}, args: options.args || {} });
        }
    });
}

function childMain () {
    const fluid = require(path.join(__dirname, 'fluid.js'));
    let init = null;
    process.on('message', async m => {
        try {
            if (m.type === 'init') {
                init = m;
                if (m.rank === 0) process.send({ type: 'commId', id: fluid.commUniqueId().toString('base64') });
            } else if (m.type === 'commId') {
                const commId = Buffer.from(m.id, 'base64');
                const tile = { rank: init.rank, world: init.world, tilesX: init.tilesX, halo: init.world > 1 ? init.halo : 0, commId };
                if (init.reach !== undefined) tile.reach = init.reach;
                if (init.linkModel !== undefined) tile.linkModel = init.linkModel;
                const sim = fluid.createFluid(Object.assign({}, init.fluid, { tile, device: init.device }));
                const work = require(init.worker);
                const value = await work(sim, { rank: init.rank, world: init.world, tilesX: init.tilesX, args: init.args, fluid });
                sim.destroy();
                process.send({ type: 'result', value: value === undefined ? null : value }, () => process.exit(0));
            }
        } catch (e) {
            process.send({ type: 'error', message: String(e && e.stack || e) }, () => process.exit(1));
        }
    });
}

if (process.argv.includes('--tile-child')) childMain();
else module.exports = { launchTiles };
