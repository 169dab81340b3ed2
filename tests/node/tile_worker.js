// worker of tests/node/run_launch_tiles.js: what one rank does with its tile (same seed on every rank -> same splats)
'use strict';
const fs = require('fs');
module.exports = async function (sim, ctx) {
    const a = ctx.args;
    sim.multipleSplats(a.randomSplats);
    sim.step(a.dt, a.steps);
    sim.sync(); sim.checkHalo();
    const names = ['velocity', 'pressure', 'divergence', 'curl', 'dye'];
    fs.writeFileSync(a.out + '.' + ctx.rank, Buffer.concat(names.map(n => Buffer.from(sim.readField(n).buffer))));
    return { rank: ctx.rank, exchanges: sim.exchangeCount(), sim: [sim.velocity.width, sim.velocity.height] };
};
